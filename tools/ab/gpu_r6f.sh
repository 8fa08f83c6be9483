# Round 6: k_conv3s_rc -- op cases (3 passes), the eval-level parity tests, the A/B in the replayed graph, the ablation table.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6f}
mkdir -p $O
export TMPDIR=/tmp
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "conv3s" > $O/conv3s_cases_$k.log 2>&1; grep "^FAILED\|passed\|failed" $O/conv3s_cases_$k.log; done
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "forward_matches_reference_golden or bitwise or plms_canonical or large_batch or medium" > $O/unet_parity.log 2>&1; grep "rel\|passed\|failed" $O/unet_parity.log | tail -12
for attrs in "" "conv3s=0"; do
  for B in 1 2 4; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_conv3s_rc_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep -v amdgpu >> $O/r06_conv3s_rc_ab.log
  done
done
cat $O/r06_conv3s_rc_ab.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b1_conv3s_rc.log; head -n 12 $O/r06_graph_ablate_b1_conv3s_rc.log
