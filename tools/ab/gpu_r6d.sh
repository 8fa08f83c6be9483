# Round 6: k_conv3s in its shipped scope -- every instantiated variant against the torch reference (3 passes: the r06 anomaly was timing dependent),
# the UNet suites, then the A/B in the replayed graph and the ablation table.    bash tools/gpu_r6d.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6d}
mkdir -p $O
export TMPDIR=/tmp
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "conv3s" > $O/conv3s_cases_$k.log 2>&1; grep "^FAILED\|passed\|failed" $O/conv3s_cases_$k.log; done
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fused.py tests/test_gpu_unet_ops.py -m gpu -q > $O/r06_gpu_unet_suites.log 2>&1; tail -n 3 $O/r06_gpu_unet_suites.log
for attrs in "" "conv3s_tw32=0,conv3s_tw16=0" "conv3s=0"; do
  for B in 1 2 4; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_conv3s_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep -v amdgpu >> $O/r06_conv3s_ab.log
  done
done
cat $O/r06_conv3s_ab.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b1_conv3s.log; head -n 16 $O/r06_graph_ablate_b1_conv3s.log
