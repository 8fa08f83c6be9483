#!/bin/bash
# r3 GPU call aa: hybrid UNet plan (large-M ResnetBlocks unfused on k_conv3_halo): parity at forced small threshold, eval-time sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3aa; mkdir -p $O
SF_UNFUSED_ROWS=512 timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "golden or sampler" > $O/tests_unet_hybrid.log 2>&1; tail -n 3 $O/tests_unet_hybrid.log
for B in 4 8 32; do for thr in 1073741824 4096 16384; do
  echo -n "B=$B SF_UNFUSED_ROWS=$thr: " | tee -a $O/sweep.log; SF_UNFUSED_ROWS=$thr timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/sweep.log
done; done
