# Round 6: 8-wave k_gemm_rows_ks, the 16x16 threshold; rocprofv3 kernel stats of the B = 32 eval loop.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6o}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "gemv or gn_act" > $O/ops.log 2>&1; tail -n 3 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 8
for B in 8 16 32; do
  timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_unet_time.log
done
cat $O/r06_unet_time.log
timeout 400 python tools/graph_ablate.py 16 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b16.log; head -n 12 $O/r06_graph_ablate_b16.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b32 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 32 6 > $GRAFT_REPO_ROOT/$O/prof_b32.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_b32 -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $O/r06_unet_eval_b32_kernel_stats.csv
head -n 25 $O/r06_unet_eval_b32_kernel_stats.csv | cut -c1-180
rm -rf $O/prof_b32
