#!/bin/bash
# r3 GPU call o: GlobalContext pooling in conv2's epilogue (SF_GCA_EPI_POOL=0: the k_gca_pool launch) -- eval time, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
for v in 1 0; do echo "== SF_GCA_EPI_POOL=$v" | tee -a $O/unet_time.log
  for B in 1 4; do SF_GCA_EPI_POOL=$v timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done; done
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -m gpu -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
