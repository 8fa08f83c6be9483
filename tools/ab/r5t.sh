cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $O
export TMPDIR=/tmp
{ timeout 100 python tools/unet_time.py 8 2>&1 | grep "eval="; timeout 100 python tools/unet_time.py 32 2>&1 | grep "eval="; } | tee $O/r05_unet_time_b8_b32.log
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-also-measured > $O/rp2.log 2>&1
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $O/r05_bench_kernel_stats.csv
