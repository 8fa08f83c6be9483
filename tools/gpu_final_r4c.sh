# Round-4 last check at HEAD: full GPU suite, the model-level tests with the IEEE-half build as the process default, the default bench line.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final4d}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_gpu_tests.log 2>&1; tail -n 3 $O/r04_gpu_tests.log
SF_OPERAND=f16 timeout 400 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q > $O/r04_gpu_tests_f16_process.log 2>&1; tail -n 2 $O/r04_gpu_tests_f16_process.log
timeout 400 python bench.py > $O/r04_bench_n1.json 2> $O/bench_n1.err
tail -n 1 $O/r04_bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'), d['roofline'].get('frac'), d['roofline'].get('traffic'), d['also_measured']['config3_B4']['ms_per_step'])"
