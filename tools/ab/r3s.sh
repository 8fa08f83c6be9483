#!/bin/bash
# r3 GPU call s: IEEE-half operand build -- UNet parity (per-module selection), eval time, and the whole process on it (SF_OPERAND=f16)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "fp16 or golden" -s > $O/tests_unet.log 2>&1; grep "fp16\|passed\|failed" $O/tests_unet.log | tail -5
SF_OPERAND=f16 timeout 200 python tools/unet_time.py 1 2>&1 | grep "sampler path" | tee $O/unet_time_f16.log
SF_OPERAND=f16 timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py tests/test_gpu_fused.py tests/test_gpu_unet_ops.py -m gpu -q -x > $O/tests_f16_process.log 2>&1; tail -n 4 $O/tests_f16_process.log
