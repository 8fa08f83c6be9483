# Round-end measurement set: bench lines, kernel traces, counters.  Everything lands in gpurun_out/<tag>/ and is copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final}
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --views-per-gpu 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n1_views4.json 2> $O/bench_v4.err
python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n1_total32.json 2> $O/bench_t32.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/r02_unet_eval_b1_timeline.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/r02_unet_eval_b1_kernel_stats.csv
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r02_ngp_microbench_kernel_stats.csv
bash tools/gpu_unet_pmc.sh ${1:-final} > $O/pmc.log 2>&1
bash tools/gpu_pmc2.sh ${1:-final} > $O/pmc2.log 2>&1
SF_TIMING_LIB=sparsefusion_amd/libsf_fused_timing.so python tools/fconv_phases.py 2>&1 | grep -v amdgpu.ids > $O/r02_fconv_phases.log
python tools/occ_eval_time.py > $O/occ_eval.log 2>&1
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
python tools/unet_time.py 4 > $O/unet_time4.log 2>&1
python tools/vae_time.py 1 > $O/vae_time.log 2>&1
for f in bench_n1 bench_n1_views4 bench_n1_total32; do tail -n 1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'))"; done
grep "^# launches" $O/r02_unet_eval_b1_timeline.txt; tail -n 2 $O/unet_time1.log $O/unet_time4.log $O/occ_eval.log $O/vae_time.log
tail -4 $O/pmc.log; tail -22 $O/pmc2.log
head -8 $O/r02_bench_kernel_stats.csv | cut -c1-120
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
timeout 240 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; tail -n 2 $O/tests_all.log
