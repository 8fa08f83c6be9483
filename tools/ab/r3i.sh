#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
for v in _nomath _noweights _noact; do
  echo "== libsf_fused_timing$v" | tee -a $O/stage_experiment2.log
  SF_TIMING_LIB=sparsefusion_amd/libsf_fused_timing$v.so timeout 120 python tools/fconv_phases.py unet_pipe_32x32_512 unet_pipe_32x32_256 unet_pipe_16x16_768 unet_pipe_8x8_1536 2>&1 | grep -v "amdgpu.ids\|(-)\|\[entry" | tee -a $O/stage_experiment2.log
done
