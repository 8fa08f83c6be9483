cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final4b}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_gpu_tests.log 2>&1; tail -n 3 $O/r04_gpu_tests.log
timeout 200 python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $O/r04_bench_n1_total32.json 2> $O/bench_t32.err; tail -c 600 $O/r04_bench_n1_total32.json
timeout 100 python tools/unet_time.py 8 2>&1 | grep "eval=" | tee $O/unet_time_b8.log
timeout 100 python tools/unet_time.py 32 2>&1 | grep "eval=" | tee -a $O/unet_time_b8.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpe -- python $GRAFT_REPO_ROOT/tools/eft_time.py 6 > $GRAFT_REPO_ROOT/$O/rpe.log 2>&1
cp $(find /tmp/rpe -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r04_eft_render_kernel_stats.csv
