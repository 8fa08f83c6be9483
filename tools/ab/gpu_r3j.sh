#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -q -m gpu -k "init_x" > $O/tests.log 2>&1; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/initx_time.py > $GRAFT_REPO_ROOT/$O/initx.log 2>&1
cd $GRAFT_REPO_ROOT; grep -E "^hot|^cold" $O/initx.log
grep -rh "k_init_x" $O/prof --include=*kernel_stats.csv | head
