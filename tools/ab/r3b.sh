#!/bin/bash
# r3 GPU call b: kernel-argument latency (tools/exp/kernarg_lat.hip) and the HIP_FORCE_DEV_KERNARG switch on the real eval
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
for v in 0 1; do
  for b in kernarg_lat kernarg_lat_pre; do
    echo "== HIP_FORCE_DEV_KERNARG=$v $b" | tee -a $O/kernarg_lat.log
    HIP_FORCE_DEV_KERNARG=$v timeout 60 tools/exp/$b 2>&1 | tee -a $O/kernarg_lat.log
  done
done
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v unet_time" | tee -a $O/unet_time.log
  HIP_FORCE_DEV_KERNARG=$v timeout 120 python tools/unet_time.py 1 2>&1 | grep -v amdgpu.ids | tee -a $O/unet_time.log
done
