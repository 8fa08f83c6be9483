# Round 4, GPU call 2: kernel trace of the B = 1 eval (plain launches) -> per-kernel durations.   bash tools/ab/r4b.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_unet_ops.py -q -x -k "glds" 2>&1 | tail -n 3
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpu -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $O/rpu.log 2>&1
cp $(find /tmp/rpu -name "*kernel_stats.csv" | head -1) $O/unet_eval_b1_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/rpu $O/unet_eval_b1_timeline.txt
tail -n 40 $O/unet_eval_b1_timeline.txt
