cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O
export TMPDIR=/tmp
{
echo "== B=4 default"; timeout 200 python tools/unet_time.py 4 2>&1 | grep "sampler path"
echo "== B=4 conv4_reduce_min_batch=4"; SF_UNET_ATTRS=conv4_reduce_min_batch=4 timeout 200 python tools/unet_time.py 4 2>&1 | grep "sampler path"
echo "== B=4 k_conv4_gn_mb<128, 1, 2> (SF_CONV4_MB_128_SPLITK=1 build)"; SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_c4mb128.so timeout 200 python tools/unet_time.py 4 2>&1 | grep "sampler path"
echo "== B=32 conv4_reduce_min_batch=4"; SF_UNET_ATTRS=conv4_reduce_min_batch=4 timeout 200 python tools/unet_time.py 32 2>&1 | grep "sampler path"
} | tee $O/unet_ab.log
