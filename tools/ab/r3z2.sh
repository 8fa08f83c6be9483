#!/bin/bash
# r3 GPU call z2: Upsample convs on the operand twin + k_conv3_halo, 32x32 layers on the halo kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -x -k "halo or glds or lds_tiled" > $O/tests_ops.log 2>&1; tail -n 2 $O/tests_ops.log
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_lpips.py tests/test_gpu_eft.py -m gpu -q > $O/tests_vae_lpips_eft.log 2>&1; tail -n 2 $O/tests_vae_lpips_eft.log
timeout 200 python tools/vae_time.py 1 2>&1 | grep -v amdgpu.ids | tee $O/vae_time.log | grep "B=1"
timeout 200 python tools/vae_layers.py > $O/vae_layers.log 2>&1; grep "==" $O/vae_layers.log
