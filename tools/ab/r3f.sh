#!/bin/bash
# r3 GPU call f: measured tile picks for the igemm layers of the UNet body; in-kernel phase stamps of the pipelined fused convs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 400 python tools/tile_sweep.py 1 2>&1 | grep -v amdgpu.ids | tee $O/tile_sweep_b1.log | tail -n 16
timeout 200 python tools/fconv_phases.py unet_pipe_8x8_1536 unet_pipe_16x16_768 unet_pipe_32x32_512 unet_pipe_32x32_256 unet_4x4_1024_s4 unet_4x4_2048_s4_gate 2>&1 | grep -v amdgpu.ids | tee $O/fconv_phases.log
