# Round 6: the driver-like bench line (with the counter passes), the per-class counter summaries at B = 1 and B = 4, the kernel-trace stats of the eval.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6e}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.err
tail -n 1 $O/r06_bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:d[k] for k in ('value','ms_per_step')}, {k:r.get(k) for k in ('frac','traffic','traffic_over_algorithmic','mfma_busy','hbm_gbs_counter','frac_whole_eval','frac_whole_eval_survey_8d_bytes','avg_launch_us','fused_conv_ms_per_eval')}, {k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also_measured',{}).items()})"
timeout 600 python tools/unet_pmc.py 1 6 $O/r06_unet_eval_b1_pmc.json > /dev/null 2> $O/pmc1.err; python -c "import json; d=json.load(open('$O/r06_unet_eval_b1_pmc.json')); print(d['fused_conv_family'], d['whole_eval'])"
timeout 600 python tools/unet_pmc.py 4 6 $O/r06_unet_eval_b4_pmc.json > /dev/null 2> $O/pmc4.err; python -c "import json; d=json.load(open('$O/r06_unet_eval_b4_pmc.json')); print(d['fused_conv_family'], d['whole_eval'])"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 20 > /dev/null 2>&1; cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r06_unet_eval_b1_kernel_stats.csv; head -5 $GRAFT_REPO_ROOT/$O/r06_unet_eval_b1_kernel_stats.csv
