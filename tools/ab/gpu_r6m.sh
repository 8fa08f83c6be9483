# Round 6 milestone: UNet GPU suites at HEAD, eval times B = 1 .. 32, the driver-like bench line, ablation tables at B = 8 / 16 / 32.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6m}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fused.py tests/test_gpu_unet_ops.py -m gpu -q > $O/r06_gpu_unet_suites.log 2>&1; tail -n 3 $O/r06_gpu_unet_suites.log
for B in 1 2 4 8 16 32; do
  timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_unet_time.log
done
cat $O/r06_unet_time.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1.json 2> $O/bench_n1.err
tail -n 1 $O/r06_bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d['roofline'].get('traffic'), d['roofline'].get('frac_whole_eval'), {k:(v.get('value'),v.get('ms_per_step'),v.get('unet_eval_ms')) for k,v in d.get('also_measured',{}).items()})"
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 14 $O/r06_graph_ablate_b32.log
timeout 300 python tools/graph_ablate.py 16 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b16.log; head -n 14 $O/r06_graph_ablate_b16.log
timeout 300 python tools/graph_ablate.py 8 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b8.log; head -n 14 $O/r06_graph_ablate_b8.log
