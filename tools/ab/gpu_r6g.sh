# Round 6: k_conv3s with the B >= 2 tiles at 16x16 / 8x8 -- op cases (3 passes), the UNet suites, eval times at B = 1, 2, 4, 8, 32, ablation at B = 4.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6g}
mkdir -p $O
export TMPDIR=/tmp
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "conv3s" > $O/conv3s_cases_$k.log 2>&1; grep "^FAILED\|passed\|failed" $O/conv3s_cases_$k.log; done
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fused.py tests/test_gpu_unet_ops.py -m gpu -q > $O/r06_gpu_unet_suites.log 2>&1; tail -n 3 $O/r06_gpu_unet_suites.log
for attrs in "" "conv3s=0"; do
  for B in 1 2 4 8 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_conv3s_final_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep -v amdgpu >> $O/r06_conv3s_final_ab.log
  done
done
cat $O/r06_conv3s_final_ab.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b4.log; head -n 14 $O/r06_graph_ablate_b4.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b1.log
