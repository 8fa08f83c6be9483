cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2h}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -8 > $O/t_fused.log
python -m pytest tests/test_gpu_unet.py -q -s 2>&1 | tail -40 > $O/t_unet.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fused.csv
python tools/unet_time.py 1 > $O/unet_time.log 2>&1
tail -n 5 $O/t_fused.log; tail -n 25 $O/t_unet.log
cat $O/unet_time.log | tail -5
grep "^# " $O/timeline_fused.txt | head -40
