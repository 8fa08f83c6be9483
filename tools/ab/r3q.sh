#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
for v in 1 0; do echo "== SF_GCA_EPI_POOL=$v" | tee -a $O/unet_time.log
  for B in 1 2 4; do SF_GCA_EPI_POOL=$v timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done; done
