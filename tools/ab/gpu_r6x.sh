# Round 6: LPIPS on IEEE-half operands by default: its tests, the experimental / e2e tests that use it, smoke, a short bench line.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6x}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lpips.py tests/test_gpu_experimental.py tests/test_gpu_e2e_distill.py -m gpu -q -s > $O/lpips.log 2>&1; grep "grad rel\|passed\|failed\|Error\|PSNR" $O/lpips.log | tail -n 16
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 4 $O/smoke.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also-measured --no-traffic > $O/bench.json 2> $O/bench.err; tail -n 1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown_ms'])"
