# r04 call 23: binned scatter tuning on the real field (2^16-row `tiled` levels): rows per bucket, entry headroom, chunks, cached-kernel run
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4p}; mkdir -p $O
export TMPDIR=/tmp
{
echo "== default (2048 rows per bucket, 4 chunks, run 32)"; timeout 100 python tools/ngp_microbench.py
for v in r10 r10h c2 r10c2 r10hc2 r10c3 r10c1; do echo "== variant $v"; SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_$v.so timeout 100 python tools/ngp_microbench.py; done
} 2>&1 | grep -v amdgpu.ids | tee $O/ngp_bin_tune.log
