# Round 6: k_gn_one, 8x8 level on k_conv3_halo_sm from B = 16, weight prefetch in k_gemm_rows_ks, unfused 4x4 linears; full UNet GPU suites.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6l}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q > $O/ops.log 2>&1; tail -n 5 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch or plms_batch8" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 20
for attrs in "" "gn_one=0" "unfused_lin_min_rows=256" "unfused_lin_min_rows=128"; do
  for B in 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_gn_one_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_gn_one_ab.log
  done
done
for attrs in "" "lds_mid_min_batch=1"; do
  for B in 1 2 4; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_gn_one_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_gn_one_ab.log
  done
done
cat $O/r06_gn_one_ab.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 24 $O/r06_graph_ablate_b32.log
timeout 400 python tools/graph_ablate.py 16 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b16.log; head -n 12 $O/r06_graph_ablate_b16.log
