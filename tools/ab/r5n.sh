cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
export TMPDIR=/tmp
for b in 1 4; do
  for at in "" "gate_t=0"; do echo "== B=$b $at"; SF_UNET_ATTRS=$at timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done
done | tee $O/unet_ab.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids | grep "full graph\|gca_" | tee $O/graph_ablate_b1.log
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fused.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
