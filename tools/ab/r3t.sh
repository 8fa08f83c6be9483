#!/bin/bash
# r3 GPU call t: k_conv_glds -- bitwise against k_conv_lds, then the SD-VAE layer table with the ring off / 3 / 4 deep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -x -k "glds or lds_tiled" > $O/tests_glds.log 2>&1; tail -n 3 $O/tests_glds.log
if grep -q "failed\|error\|Timeout" $O/tests_glds.log; then exit 1; fi
for d in 0 3 4; do SF_CONV_GLDS=$d timeout 200 python tools/vae_layers.py > $O/vae_layers_glds$d.log 2>&1; grep "==" $O/vae_layers_glds$d.log; done
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_lpips.py tests/test_gpu_eft.py -m gpu -q > $O/tests_vae_lpips_eft.log 2>&1; tail -n 3 $O/tests_vae_lpips_eft.log
