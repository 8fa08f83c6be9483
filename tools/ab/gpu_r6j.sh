# Round 6: k_gemm_rows_ks + thresholds
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6j}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "gemv" > $O/ops.log 2>&1; tail -n 5 $O/ops.log
for attrs in "" "unfused_min_rows_4=128"; do
  for B in 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_mid_ab2.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_mid_ab2.log
  done
done
cat $O/r06_mid_ab2.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 14 $O/r06_graph_ablate_b32.log
