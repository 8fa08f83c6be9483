cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_dbg8192.so timeout 300 python tools/exp/conv3s_lds_dump.py conv3s_8x8_512_pool 2>&1 | grep -v amdgpu
