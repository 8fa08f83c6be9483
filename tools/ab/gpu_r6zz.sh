# Round 6: the driver-like bench line once more (the counter pass's kernel-name filter now knows k_lin4_attn).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6zz}
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1_final.json 2> $O/bench_n1.err
tail -n 1 $O/r06_bench_n1_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:d[k] for k in ('value','ms_per_step')}, r['frac'], r['traffic'], r['traffic_over_algorithmic'], r['counters']['dispatches_per_eval'], r['frac_whole_eval_survey_8d_bytes'], {k:(v.get('value'),v.get('ms_per_step'),v.get('unet_eval_ms')) for k,v in d.get('also_measured',{}).items()})"
