#!/bin/bash
# r3 GPU call y: k_conv3_halo parity + time against k_conv_lds / k_conv_glds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -x -k "halo or glds or lds_tiled" > $O/tests.log 2>&1; tail -n 3 $O/tests.log
timeout 120 python tools/conv_time.py 1 4 6 7 2>&1 | grep -v amdgpu.ids | tee $O/conv_time.log
