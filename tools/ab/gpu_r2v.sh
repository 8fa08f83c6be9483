# Round-2 session-2 call 4: pipe kernel with deeper staging prefetch + slot loads first + pipe pairs; GN_SELF reorder.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2v}
mkdir -p $O
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/sparsefusion_amd
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -q -x 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
for p in "1 1" "1 0" "0 1"; do
  set -- $p
  echo "== SF_PIPE=$1 SF_PAIR=$2" >> $O/unet_time.log
  SF_PIPE=$1 SF_PAIR=$2 timeout 300 python tools/unet_time.py 1 2>&1 | tail -2 >> $O/unet_time.log
done
cat $O/unet_time.log
CASES="unet_4x4_1024_s4 unet_4x4_2048_s4_gate unet_pipe_8x8_1536 unet_pipe_16x16_768 unet_pipe_32x32_512 unet_pipe_32x32_256 unet_ln_qkv_lazy"
SF_TIMING_LIB=$L/libsf_fused_timing.so timeout 300 python tools/fconv_phases.py $CASES 2>&1 | grep -v amdgpu.ids > $O/phases.log
cat $O/phases.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/unet_eval_b1_timeline.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/unet_eval_b1_kernel_stats.csv
grep "^#" $O/unet_eval_b1_timeline.txt | head -34
