# Round 6: k_lin4_attn (the attention-prologue projection of the 16-token map on its own kernel): fused cases, UNet suites, eval times (the general kernel = conv4=0 is no clean A/B: compare with profiles/r06_unet_time.log), ablation at B = 1.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6z}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "attention" > $O/fused.log 2>&1; tail -n 3 $O/fused.log
for k in 1 2; do for B in 1 2 4 8; do timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_unet_time.log; done; done; cat $O/r06_unet_time.log
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b1.log; grep "attn\|full graph" $O/r06_graph_ablate_b1.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b4.log; grep "attn\|full graph" $O/r06_graph_ablate_b4.log
timeout 1500 python -m pytest tests/test_gpu_unet.py -m gpu -q > $O/unet.log 2>&1; tail -n 4 $O/unet.log
