set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -25 > gpurun_out/r2a/t_fused.log
python -m pytest tests/test_gpu_unet.py -q 2>&1 | tail -25 > gpurun_out/r2a/t_unet.log
python tools/unet_profile.py 1 1024 > gpurun_out/r2a/prof_fused.log 2>&1
SF_UNET_FUSED=0 python tools/unet_profile.py 1 1024 > gpurun_out/r2a/prof_unfused.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/gpurun_out/r2a/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 gpurun_out/r2a/timeline_fused.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) gpurun_out/r2a/kernel_stats_fused.csv
tail -5 gpurun_out/r2a/t_fused.log gpurun_out/r2a/t_unet.log
tail -12 gpurun_out/r2a/prof_fused.log
tail -30 gpurun_out/r2a/timeline_fused.txt
