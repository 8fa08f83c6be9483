# r04 binned scatter tuning on the real field (2^16-row `tiled` levels): library variants by SF_HIP_LIB
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4p}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ngp.py -m gpu -q -k "binned or isolated or full_size or golden" 2>&1 | tail -n 3
{
echo "== default"; timeout 100 python tools/ngp_microbench.py
for v in $VARIANTS; do echo "== variant $v"; SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_$v.so timeout 100 python tools/ngp_microbench.py; done
} 2>&1 | grep -v amdgpu.ids | tee $O/ngp_bin_tune.log
