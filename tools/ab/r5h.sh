cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
echo "== staged stores (product)"; timeout 300 python tools/ngp_microbench.py 2>&1 | grep render
echo "== r04 direct stores";      SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_nostage.so timeout 300 python tools/ngp_microbench.py 2>&1 | grep render
done | tee $O/ngp_stage_ab.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r05_ngp_microbench_kernel_stats.csv
SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_nostage.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn0 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn0.log 2>&1
cp $(find /tmp/rpn0 -name "*kernel_stats.csv" | head -1) $O/r05_ngp_microbench_kernel_stats_direct_stores.csv
cd $GRAFT_REPO_ROOT
head -8 $O/r05_ngp_microbench_kernel_stats.csv | cut -c1-150; echo; head -8 $O/r05_ngp_microbench_kernel_stats_direct_stores.csv | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_ngp.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_ngp.log
