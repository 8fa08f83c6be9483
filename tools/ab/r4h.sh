cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4h}; mkdir -p $O
timeout 300 python tools/fconv_phases.py unet_ln_ff1_1024 unet_ln_ff2_2048_plain unet_ln_qkv_1024 attn_self unet_4x4_1024_s4 2>&1 | grep -v amdgpu.ids | tee $O/phases.log
