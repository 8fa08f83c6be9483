#!/bin/bash
# r3 GPU call v: L2-channel spread of the A loads (per-tile rotation of the 64-channel chunk order)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; mkdir -p $O
for n in 7 8; do echo "== experiment $n" | tee -a $O/conv_time.log; SF_HIP_LIB=$PWD/sparsefusion_amd/libsparsefusion_hip_glds_x$n.so timeout 120 python tools/conv_time.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_time.log; done
