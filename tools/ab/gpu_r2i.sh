cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2i}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -8 > $O/t_fused.log
python -m pytest tests/test_gpu_unet.py -q -k "forward_matches or plms_sampler" 2>&1 | tail -8 > $O/t_unet.log
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
python tools/unet_time.py 4 > $O/unet_time4.log 2>&1
python tools/fconv_phases.py unet_ln_ff2_2048 unet_ln_qkv_lazy unet_4x4_1024_s4 > $O/phases.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
tail -n 3 $O/t_fused.log; tail -n 3 $O/t_unet.log
tail -n 3 $O/unet_time1.log $O/unet_time4.log
cat $O/phases.log
cat $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','breakdown_ms')})"
tail -3 $O/bench.err
grep "^# " $O/timeline_fused.txt | head -16
