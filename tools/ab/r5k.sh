cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "lin4 or conv4 or layernorm or unet_ln" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_fused.log
for b in 1 4; do
  for c in 1 0; do echo "== B=$b conv4=$c"; SF_UNET_ATTRS=conv4=$c timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done
done | tee $O/unet_conv4_ab.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids | head -14 | tee $O/graph_ablate_b1.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_unet.log
