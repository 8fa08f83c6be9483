cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
for b in 2 4 8; do
  for at in "" "conv4_batch=0"; do echo "== B=$b $at"; SF_UNET_ATTRS=$at timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done
done | tee $O/unet_ab.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu.ids | grep "full graph\|4x4" | tee $O/graph_ablate_b4.log
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fused.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
