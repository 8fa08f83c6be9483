cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2m}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -4 > $O/t_fused.log
python tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 unet_16x16_768 unet_32x32_res_conv > $O/phases.log 2>&1
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
tail -n 2 $O/t_fused.log
grep -A3 "^unet" $O/phases.log | grep "unet\|staging\|prefetch"
tail -n 2 $O/unet_time1.log
