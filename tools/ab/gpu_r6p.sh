# Round 6: 8-wave k_gemm_rows_ks (LDS fixed), the 16x16 threshold, the B = 1 tile for the (conv1 || res_conv) pairs of B >= 2; rocprofv3 kernel stats of the B = 32 / B = 1 eval loops.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6p}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "gemv or gn_act" > $O/ops.log 2>&1; tail -n 3 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 8
for attrs in "" "rc_small_tiles=32" "rc_small_tiles=48"; do
  for B in 2 4 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_rc_small_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_rc_small_ab.log
  done
done
cat $O/r06_rc_small_ab.log
timeout 400 python tools/graph_ablate.py 16 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b16.log; head -n 12 $O/r06_graph_ablate_b16.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks32 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 32 6 > /dev/null 2>&1; cp $(find /tmp/ks32 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r06_unet_eval_b32_kernel_stats.csv
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 20 > /dev/null 2>&1; cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r06_unet_eval_b1_kernel_stats.csv
cd $GRAFT_REPO_ROOT; head -n 16 $O/r06_unet_eval_b32_kernel_stats.csv | cut -c1-150
