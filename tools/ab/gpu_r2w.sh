# Round-2 session-2 call 5: full GPU suite + bench on the current tree.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2w}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/unet_time.py 1 2>&1 | tail -2 > $O/unet_time.log
cat $O/unet_time.log
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/tests.log
cat $O/tests.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -n 1 $O/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms')); print(d['roofline'])"
tail -3 $O/bench_n1.err
