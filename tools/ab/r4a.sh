# Round 4, GPU call 1: parity of the r04 launch fusions + eval time A/B.   bash tools/ab/r4a.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4a}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -q -x 2>&1 | tail -n 15 > $O/tests_unet.log; tail -n 3 $O/tests_unet.log
timeout 600 python -m pytest tests/test_gpu_unet_ops.py tests/test_gpu_ngp.py tests/test_gpu_bench_multirank.py -q -x -k "glds or shuffle or field_cache or render or spawns" 2>&1 | tail -n 8 > $O/tests_misc.log; tail -n 3 $O/tests_misc.log
for B in 1 4; do
  echo "== B=$B r04 plan" | tee -a $O/unet_time.log; timeout 120 python tools/unet_time.py $B 2>&1 | grep "eval=" | tee -a $O/unet_time.log
  echo "== B=$B r03 plan (attn_in_out_proj=0, producer_slots=0)" | tee -a $O/unet_time.log
  SF_UNET_ATTRS="attn_in_out_proj=0,producer_slots=0" timeout 120 python tools/unet_time.py $B 2>&1 | grep "eval=" | tee -a $O/unet_time.log
done
for v in ntpipe0 prio1 prio2; do
  echo "== B=1 variant $v" | tee -a $O/unet_time.log
  SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_$v.so timeout 120 python tools/unet_time.py 1 2>&1 | grep "sampler path" | tee -a $O/unet_time.log
done
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 2500 $O/bench_n1.json
