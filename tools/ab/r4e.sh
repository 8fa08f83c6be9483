cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4e}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -q -x 2>&1 | tail -n 15 > $O/tests_unet.log; tail -n 3 $O/tests_unet.log
timeout 600 python -m pytest tests/test_gpu_eft.py tests/test_gpu_bench_multirank.py -q -x 2>&1 | tail -n 15 > $O/tests_misc.log; tail -n 3 $O/tests_misc.log
for B in 1 4; do
  echo "== B=$B r04 plan" | tee -a $O/unet_time.log; timeout 120 python tools/unet_time.py $B 2>&1 | grep "eval=" | tee -a $O/unet_time.log
  echo "== B=$B gca_one_launch=0" | tee -a $O/unet_time.log
  SF_UNET_ATTRS="gca_one_launch=0" timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
done
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids | tee $O/graph_ablate_b1.log
timeout 300 python bench.py --config 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_c2.err; tail -c 1200 $O/bench_config2.json
