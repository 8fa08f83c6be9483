cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4i}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 40 > $O/gpu_tests.log; tail -n 5 $O/gpu_tests.log
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1.json
