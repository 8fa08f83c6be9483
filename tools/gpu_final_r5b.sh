# Round-5 closing set (after k_conv4_gn_mb / the planner's B >= 4 rules): the driver-like bench line, then the full GPU suite.
#   bash tools/gpu_final_r5b.sh <tag>      results in gpurun_out/<tag>/ (copied to profiles/ as r05_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final5c}
mkdir -p $O
export TMPDIR=/tmp
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1_final.json 2> $O/bench_n1.err
tail -n 1 $O/r05_bench_n1_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('traffic'), d['roofline'].get('frac_whole_eval'), {k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also_measured',{}).items()})"
timeout 600 python -m pytest tests -m gpu -q > $O/r05_gpu_tests.log 2>&1; tail -n 3 $O/r05_gpu_tests.log
