#!/bin/bash
# r3 GPU call g: where does the staging time of the pipelined fused convs go?  (timing builds without SiLU / without arithmetic)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
for v in "" _nosilu _nomath; do
  echo "== libsf_fused_timing$v" | tee -a $O/stage_experiment.log
  SF_TIMING_LIB=sparsefusion_amd/libsf_fused_timing$v.so timeout 120 python tools/fconv_phases.py unet_pipe_32x32_512 unet_pipe_32x32_256 unet_pipe_16x16_768 2>&1 | grep -v amdgpu.ids | tee -a $O/stage_experiment.log
done
