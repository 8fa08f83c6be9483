cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $O
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_earlybias.so
{
for rep in 1 2; do
echo "== B=1 product"; timeout 100 python tools/unet_time.py 1 2>&1 | grep "sampler path"
echo "== B=1 SF_EARLY_BIAS=1"; SF_HIP_LIB=$V timeout 100 python tools/unet_time.py 1 2>&1 | grep "sampler path"
done
echo "== B=4 product"; timeout 100 python tools/unet_time.py 4 2>&1 | grep "sampler path"
echo "== B=4 SF_EARLY_BIAS=1"; SF_HIP_LIB=$V timeout 100 python tools/unet_time.py 4 2>&1 | grep "sampler path"
} | tee $O/unet_ab.log
SF_HIP_LIB=$V timeout 300 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet_ops.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests_variant.log
