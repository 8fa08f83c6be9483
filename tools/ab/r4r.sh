# r04 call 30: NGP tests at HEAD (incl. the large-hash-map fallback test), NGP microbench, default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4r}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_occ_render.py tests/test_gpu_e2e_distill.py -m gpu -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
timeout 100 python tools/ngp_microbench.py 2>&1 | grep render
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -n 1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'), d['roofline'].get('frac'))"
