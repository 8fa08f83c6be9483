#!/bin/bash
# r3 GPU call k: 32-pixel tiles of the 8x8 / 16x16 maps from B = 4 on (SF_BIG_TILE_B=999: off) -- B = 4 / 32 eval time, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
for v in 4 999; do echo "== SF_BIG_TILE_B=$v" | tee -a $O/unet_time.log
  for B in 4 32; do SF_BIG_TILE_B=$v timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done; done
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -m gpu -q > $O/tests.log 2>&1; tail -n 3 $O/tests.log
