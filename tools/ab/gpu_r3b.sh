cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3b}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py tests/test_gpu_ngp.py tests/test_gpu_unet_ops.py -q -x 2>&1 | tail -3 | tee $O/tests.log
for p in 1 0; do echo "== SF_POOLNET=$p"; SF_POOLNET=$p python tools/unet_time.py 1 2>&1 | tail -2; done | tee $O/unet_time.log
for t in 8 4 16; do echo "== scatter run (compiled 8) cutoff sweep"; done > /dev/null
for cfg in "1024 640 1" "1024 100000 1" "1024 1100 1" "512 640 1"; do set -- $cfg; echo "== scatter threads=$1 cutoff=$2 split=$3"; SF_SC_THREADS=$1 SF_SC_CUTOFF=$2 SF_SC_SPLIT=$3 python tools/ngp_microbench.py 2>&1 | tail -2; done | tee $O/scatter.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/ngp_microbench_kernel_stats.csv
head -6 $O/ngp_microbench_kernel_stats.csv | cut -c1-120
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'))" | tee $O/bench.log
