#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
for B in 4 32; do timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done
echo "== B=2: threshold 2 vs 4" | tee -a $O/unet_time.log
for v in 2 4; do SF_BIG_TILE_B=$v timeout 200 python tools/unet_time.py 2 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -m gpu -q > $O/tests.log 2>&1; tail -n 3 $O/tests.log
