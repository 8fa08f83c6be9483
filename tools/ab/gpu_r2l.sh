cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2l}
mkdir -p $O
for U in 8 24; do SF_TIMING_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsf_fused_timing_u$U.so python tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 unet_16x16_768 unet_32x32_res_conv > $O/phases_u$U.log 2>&1; done
python tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 unet_16x16_768 unet_32x32_res_conv > $O/phases_u16.log 2>&1
for U in 8 16 24; do echo "== U=$U"; grep -A3 "^unet" $O/phases_u$U.log | grep "unet\|staging"; done
