#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/fconv_phases.py unet_pipe_32x32_256 unet_pipe_pool_32x32_256 unet_pipe_pool_8x8_1024 2>&1 | grep -v "amdgpu.ids\|(-)\|\[entry" | tee $O/phases.log
cd /tmp
for v in 1 0; do
  SF_GCA_EPI_POOL=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$v -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > /dev/null 2>&1
  cp $(find /tmp/rp$v -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/stats_pool$v.csv
done
