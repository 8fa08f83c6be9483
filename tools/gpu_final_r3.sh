# Round-3 measurement set: bench lines (configs 1 / 2 / 3, strong-scaling N = 1 baseline), kernel traces, counter passes.
# Everything lands in gpurun_out/<tag>/ and is copied to profiles/ afterwards.   bash tools/gpu_final_r3.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final3}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py > $O/r03_bench_n1.json 2> $O/bench_n1.err
timeout 200 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_views4.json 2> $O/bench_v4.err
timeout 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_n1_config2.json 2> $O/bench_c2.err
timeout 200 python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_total32.json 2> $O/bench_t32.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/r03_unet_eval_b1_timeline.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/r03_unet_eval_b1_kernel_stats.csv
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r03_ngp_microbench_kernel_stats.csv
python tools/trace_timeline.py /tmp/rpn /dev/null > /dev/null 2>&1
timeout 400 bash tools/gpu_unet_pmc_r3.sh ${1:-final3} > $O/pmc.log 2>&1
timeout 700 bash tools/gpu_pmc2_r3.sh ${1:-final3} > $O/pmc2.log 2>&1
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
python tools/unet_time.py 4 > $O/unet_time4.log 2>&1
python tools/vae_time.py 1 2>&1 | grep "^B=" > $O/vae_time.log
python tools/occ_eval_time.py > $O/occ_eval.log 2>&1
for f in r03_bench_n1 r03_bench_n1_views4 r03_bench_n1_config2 r03_bench_n1_total32; do tail -n 1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'))"; done
grep "^# launches" $O/r03_unet_eval_b1_timeline.txt; tail -n 2 $O/unet_time1.log $O/unet_time4.log $O/occ_eval.log $O/vae_time.log
tail -4 $O/pmc.log; tail -30 $O/pmc2.log
