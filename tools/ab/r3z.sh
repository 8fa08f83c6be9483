#!/bin/bash
# r3 GPU call z: the SD-VAE / LPIPS / EFT plans on the new conv kernels (defaults), the layer table, wall times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_lpips.py tests/test_gpu_eft.py -m gpu -q > $O/tests_vae_lpips_eft.log 2>&1; tail -n 3 $O/tests_vae_lpips_eft.log
timeout 200 python tools/vae_layers.py > $O/vae_layers.log 2>&1; grep "==" $O/vae_layers.log
timeout 200 python tools/vae_time.py 1 2>&1 | grep -v amdgpu.ids | tee $O/vae_time.log | head -20
SF_CONV_HALO=0 SF_CONV_GLDS=0 timeout 200 python tools/vae_time.py 1 2>&1 | grep -v amdgpu.ids | tee $O/vae_time_old.log | head -8
