cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in dbg8 dbg16; do
echo "=== $v"
SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_$v.so timeout 300 python tools/exp/conv3s_kmask.py conv3s_b2_32x32_256_strip2_wn2 2>&1 | grep "sub-chunk"
done
