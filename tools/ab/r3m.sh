#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
for B in 1 4; do timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/unet_time.log; done
timeout 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -n 4 $O/tests.log
