cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4m}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x -k "wm4" 2>&1 | tail -n 3
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "lazy_consumers or plan_switches" 2>&1 | tail -n 3
for B in 4 8; do
  echo "== B=$B 64-pixel tiles at 16x16 / 8x8" | tee -a $O/unet_time.log; timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
  echo "== B=$B huge_tile_min_batch=999" | tee -a $O/unet_time.log
  SF_UNET_ATTRS="huge_tile_min_batch=999" timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
done
