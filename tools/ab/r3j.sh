#!/bin/bash
# r3 GPU call j: cooperative L2 weight prefetch in the pipelined convs: phases + whole eval A/B (default vs -DSF_W_PREFETCH=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 120 python tools/fconv_phases.py unet_pipe_32x32_512 unet_pipe_32x32_256 unet_pipe_16x16_768 unet_pipe_8x8_1536 2>&1 | grep -v "amdgpu.ids\|(-)\|\[entry" | tee $O/phases_prefetch.log
for lib in "" sparsefusion_amd/libsparsefusion_hip_nopf.so; do
  echo "== SF_HIP_LIB=$lib" | tee -a $O/unet_time.log
  for B in 1 4; do SF_HIP_LIB=$lib timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done
done
