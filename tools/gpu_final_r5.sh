# Round-5 measurement set: full GPU suite, the bench lines (default, configs[2] / [3] / [4], 32 views on one GPU), kernel trace of the step.
#   bash tools/gpu_final_r5.sh <tag>      results in gpurun_out/<tag>/ (copied to profiles/ as r05_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final5}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r05_gpu_tests.log 2>&1; tail -n 3 $O/r05_gpu_tests.log
timeout 400 python bench.py > $O/r05_bench_n1.json 2> $O/bench_n1.err
timeout 200 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r05_bench_n1_views4.json 2> $O/bench_v4.err
timeout 200 python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/r05_bench_n1_total32.json 2> $O/bench_t32.err
timeout 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r05_bench_n1_config2.json 2> $O/bench_c2.err
timeout 200 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/r05_bench_n1_config4.json 2> $O/bench_c4.err
for f in r05_bench_n1 r05_bench_n1_views4 r05_bench_n1_total32 r05_bench_n1_config2 r05_bench_n1_config4; do tail -n 1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'), d['roofline'].get('frac'), d['roofline'].get('traffic'))"; done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-also-measured > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r05_bench_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpu -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rpu.log 2>&1
cp $(find /tmp/rpu -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r05_unet_eval_b1_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/rpu $GRAFT_REPO_ROOT/$O/r05_unet_eval_b1_timeline.txt
cd $GRAFT_REPO_ROOT
timeout 200 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids > $O/r05_graph_ablate_b1.log
SF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/r05_bench_2ranks_gloo_dryrun.json 2> $O/bench_2r.err
timeout 200 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu.ids > $O/r05_graph_ablate_b4.log
timeout 100 python tools/unet_time.py 8 2>&1 | grep "eval=" | tee $O/unet_time_b8.log
timeout 100 python tools/unet_time.py 32 2>&1 | grep "eval=" | tee -a $O/unet_time_b8.log
