# Trimmed round-end set (no counter passes): bench lines + kernel traces, most important first; stops launching new work at the deadline.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final2}
DEADLINE=$(( $(date +%s) + ${2:-165} ))
left() { echo $(( DEADLINE - $(date +%s) )); }
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -n 1 $O/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_n1', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'))"
cd /tmp
if [ $(left) -gt 35 ]; then
  timeout $(left) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
  (cd $GRAFT_REPO_ROOT; python tools/trace_timeline.py /tmp/rp1 $O/r02_unet_eval_b1_timeline.txt; cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/r02_unet_eval_b1_kernel_stats.csv; grep "^# launches" $O/r02_unet_eval_b1_timeline.txt)
fi
if [ $(left) -gt 20 ]; then
  timeout $(left) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
  (cd $GRAFT_REPO_ROOT; cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r02_ngp_microbench_kernel_stats.csv; grep render $O/rpn.log)
fi
if [ $(left) -gt 40 ]; then
  timeout $(left) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
  (cd $GRAFT_REPO_ROOT; cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv)
fi
cd $GRAFT_REPO_ROOT
if [ $(left) -gt 30 ]; then
  timeout $(left) python bench.py --views-per-gpu 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n1_views4.json 2> $O/bench_v4.err
  tail -n 1 $O/bench_n1_views4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views4', {k:d[k] for k in ('value','ms_per_step','scaling')})"
fi
if [ $(left) -gt 30 ]; then
  timeout $(left) python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n1_total32.json 2> $O/bench_t32.err
  tail -n 1 $O/bench_n1_total32.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('total32', {k:d[k] for k in ('value','ms_per_step','scaling')})"
fi
grep -hE "k_gemm_rows|k_gemv|k_ngp_composite" $O/r02_bench_kernel_stats.csv $O/r02_ngp_microbench_kernel_stats.csv 2>/dev/null | cut -c1-160
echo "left at exit: $(left) s"
