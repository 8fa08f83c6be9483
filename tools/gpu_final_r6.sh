# Round-6 closing set: the driver-like bench line, the full GPU suite, eval times B = 1 .. 32, ablation tables, rocprofv3 kernel stats of the bench.
#   bash tools/gpu_final_r6.sh <tag>      results in gpurun_out/<tag>/ (copied to profiles/ as r06_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final6}
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1_final.json 2> $O/bench_n1.err
tail -n 1 $O/r06_bench_n1_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('traffic'), d['roofline'].get('frac_whole_eval_survey_8d_bytes'), {k:(v.get('value'),v.get('ms_per_step'),v.get('unet_eval_ms')) for k,v in d.get('also_measured',{}).items()})"
for B in 1 2 4 8 16 32; do
  timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_unet_time_final.log
done
cat $O/r06_unet_time_final.log
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_gpu_tests.log 2>&1; tail -n 4 $O/r06_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -n 4 $O/r06_smoke.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b4_final.log; head -n 8 $O/r06_graph_ablate_b4_final.log
timeout 300 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32_final.log; head -n 8 $O/r06_graph_ablate_b32_final.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpb -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-also-measured > $GRAFT_REPO_ROOT/$O/rpb.log 2>&1; cp $(find /tmp/rpb -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r06_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT; head -n 12 $O/r06_bench_kernel_stats.csv | cut -c1-150
