#!/bin/bash
# r3 GPU call u: where does a k_conv_glds stage go?  measurement builds of csrc/conv_glds.h (SF_GLDS_EXPERIMENT)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
echo "== product" | tee $O/conv_time.log
timeout 120 python tools/conv_time.py 1 3 4 2>&1 | tee -a $O/conv_time.log
for n in 1 2 3 4 5 6; do echo "== experiment $n" | tee -a $O/conv_time.log; SF_HIP_LIB=$PWD/sparsefusion_amd/libsparsefusion_hip_glds_x$n.so timeout 120 python tools/conv_time.py 4 2>&1 | tee -a $O/conv_time.log; done
