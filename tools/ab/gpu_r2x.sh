cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2x}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_eft.py tests/test_gpu_e2e_distill.py tests/test_gpu_bench_multirank.py -q -x -s 2>&1 | grep -v "^$" | tail -15 > $O/tests.log
cat $O/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
tail -n 1 $O/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'))"
tail -3 $O/bench_n1.err
