#!/bin/bash
# r3 GPU call h: XCD tile map by fabric bytes (auto) vs the r02 map (SF_XCD_R=1) vs forced 2 / 4 on the real eval; fused-kernel parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
for r in "" 1 2 4; do echo "== SF_XCD_R=$r" | tee -a $O/unet_time.log; SF_XCD_R=$r timeout 120 python tools/unet_time.py 1 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done
echo "== B=4 auto / R=1" | tee -a $O/unet_time.log
timeout 120 python tools/unet_time.py 4 2>&1 | grep "sampler path" | tee -a $O/unet_time.log
SF_XCD_R=1 timeout 120 python tools/unet_time.py 4 2>&1 | grep "sampler path" | tee -a $O/unet_time.log
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -m gpu -q > $O/tests.log 2>&1; tail -n 3 $O/tests.log
