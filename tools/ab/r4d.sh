cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4d}; mkdir -p $O
timeout 300 python tools/fconv_phases.py unet_ln_ff2_2048 unet_ln_qkv_lazy unet_4x4_1024_s4 unet_4x4_2048_s4_gate attn_self attn_cross unet_pipe_32x32_256 unet_pipe_8x8_1536 2>&1 | grep -v amdgpu.ids | tee $O/phases.log
timeout 200 python tools/graph_ablate.py 1 2>&1 | grep "attn\|full" | tee $O/ablate_attn.log
