cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2n}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/t_all.log
python -m pytest tests/test_gpu_e2e_distill.py tests/test_gpu_unet.py -q -s -k "distillation or canonical_config" 2>&1 | grep -i "after\|plms canonical\|passed\|failed" > $O/t_e2e.log
bash tools/gpu_unet_pmc.sh ${1:-r2n} > $O/pmc.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/r02_unet_eval_b1_kernel_stats.csv
python bench.py --steps 4 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -n 6 $O/t_all.log; cat $O/t_e2e.log
tail -n 22 $O/pmc.log
grep "^# launches" $O/timeline_fused.txt
tail -n 1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','breakdown_ms')}); print(d['roofline']); print(d.get('cpu_baseline'))"
tail -3 $O/bench.err
