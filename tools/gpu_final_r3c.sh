# Round-3 last measurement set (after the NGP field cache): full GPU suite, the bench lines, the kernel trace of the step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final3d}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; tail -n 2 $O/r03_gpu_tests.log
timeout 300 python bench.py > $O/r03_bench_n1.json 2> $O/bench_n1.err
timeout 200 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_views4.json 2> $O/bench_v4.err
timeout 200 python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_total32.json 2> $O/bench_t32.err
for f in r03_bench_n1 r03_bench_n1_views4 r03_bench_n1_total32; do tail -n 1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'))"; done
timeout 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_n1_config2.json 2> $O/bench_c2.err
tail -n 1 $O/r03_bench_n1_config2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'])"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r03_bench_kernel_stats.csv
