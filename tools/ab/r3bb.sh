#!/bin/bash
# r3 GPU call bb: LPIPS with conv -> conv twins (forward on k_conv3_halo): parity + per-plan time, A/B against the fp32 reads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bb; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_lpips.py tests/test_gpu_losses.py -m gpu -q > $O/tests_lpips.log 2>&1; tail -n 2 $O/tests_lpips.log
timeout 120 python tools/lpips_time.py 2>&1 | grep "lds_min=96" | tee $O/lpips_time.log
SF_LPIPS_TWIN=0 timeout 120 python tools/lpips_time.py 2>&1 | grep "lds_min=96" | sed 's/^/twin=0 /' | tee -a $O/lpips_time.log
SF_LPIPS_TWIN=0 SF_CONV_HALO=0 SF_CONV_GLDS=0 timeout 120 python tools/lpips_time.py 2>&1 | grep "lds_min=96" | sed 's/^/old kernels /' | tee -a $O/lpips_time.log
