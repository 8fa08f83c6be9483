set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2b}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -15 > $O/t_fused.log
python -m pytest tests/test_gpu_unet.py -q -k "forward_matches or lazy_consumers" 2>&1 | tail -15 > $O/t_unet.log
python tools/unet_profile.py 1 1024 > $O/prof_fused.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fused.csv
tail -n 4 $O/t_fused.log; tail -n 4 $O/t_unet.log
head -16 $O/prof_fused.log
tail -n 30 $O/timeline_fused.txt
