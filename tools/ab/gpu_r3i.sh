#!/bin/bash
# init-x as MFMA-from-LDS: parity + eval time + kernel time
O=gpurun_out/r3i; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -q -m gpu -k "init_x" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "plan or sampler or eval" > $O/tests_unet.log 2>&1; tail -3 $O/tests_unet.log
SF_INITX=1 timeout 200 python tools/unet_time.py > $O/ut1.log 2>&1; tail -2 $O/ut1.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep -rh "k_init_x" $O/prof2 --include=*kernel_stats.csv | head -3
