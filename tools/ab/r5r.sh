cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "conv4" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_fused.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -s -k "r05_specialised or lazy_consumers" 2>&1 | grep -v amdgpu.ids | grep "rel L2\|passed\|failed\|Error\|error" | tee $O/tests_unet.log
timeout 600 python -m pytest tests/test_gpu_bench_multirank.py -x -q -k "single_rank" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/tests_rccl.log
for b in 2 4 32; do
  for at in "" "conv4_mb=0"; do echo "== B=$b $at"; SF_UNET_ATTRS=$at timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done
done | tee $O/unet_ab.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu.ids | grep "full graph\|4x4" | tee $O/graph_ablate_b4.log
