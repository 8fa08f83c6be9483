cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_e2e_distill.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
for b in 3 4; do timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done | tee $O/unet_time.log
