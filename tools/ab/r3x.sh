#!/bin/bash
# r3 GPU call x: k_conv_glds with the LDS-transposed float4 epilogue: parity, time, and the no-epilogue bound (x9)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -x -k "glds or lds_tiled" > $O/tests_glds.log 2>&1; tail -n 3 $O/tests_glds.log
echo "== product" | tee $O/conv_time.log
timeout 120 python tools/conv_time.py 1 3 4 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_time.log
echo "== experiment 9 (no epilogue)" | tee -a $O/conv_time.log
SF_HIP_LIB=$PWD/sparsefusion_amd/libsparsefusion_hip_glds_x9.so timeout 120 python tools/conv_time.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_time.log
