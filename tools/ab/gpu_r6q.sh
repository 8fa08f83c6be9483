# Round 6: the un-normalised k_conv4_gn (merged last-Downsample conv at B = 1): fused cases, UNet suites, B = 1 eval time A/B, ablation at B = 1.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6q}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "conv4" > $O/fused.log 2>&1; tail -n 3 $O/fused.log
for attrs in "" "merged_down_conv4=0"; do
  for k in 1 2; do
    echo "== SF_UNET_ATTRS=$attrs" >> $O/r06_merged_down_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py 1 2>&1 | grep "sampler path" >> $O/r06_merged_down_ab.log
  done
done
cat $O/r06_merged_down_ab.log
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_plms.py -m gpu -q > $O/unet.log 2>&1; tail -n 4 $O/unet.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b1_final.log; head -n 30 $O/r06_graph_ablate_b1_final.log
