cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2j}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -4 > $O/t_fused.log
python -m pytest tests/test_gpu_unet.py -q -k "forward_matches or plms_sampler or fast_path" 2>&1 | tail -6 > $O/t_unet.log
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fused.csv
tail -n 3 $O/t_fused.log; tail -n 3 $O/t_unet.log
tail -n 2 $O/unet_time1.log
grep "^# " $O/timeline_fused.txt | head -30
