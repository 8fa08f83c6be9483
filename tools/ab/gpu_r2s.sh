# Round-2 session-2 call 1: parity of the reworked fused prologue / pair launches / scatter split, then A/B timing sweeps.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2s}
mkdir -p $O
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/sparsefusion_amd
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py tests/test_gpu_ngp.py -q -x 2>&1 | tail -6 > $O/tests.log
cat $O/tests.log
for v in base r0 r2 u8 u2 nt; do
  if [ $v = base ]; then lib=$L/libsparsefusion_hip.so; else lib=$L/libsparsefusion_hip_$v.so; fi
  echo "== variant $v" >> $O/unet_time.log
  SF_HIP_LIB=$lib timeout 300 python tools/unet_time.py 1 2>&1 | tail -2 >> $O/unet_time.log
done
echo "== base, SF_PAIR=0" >> $O/unet_time.log
SF_PAIR=0 timeout 300 python tools/unet_time.py 1 2>&1 | tail -2 >> $O/unet_time.log
cat $O/unet_time.log
CASES="unet_4x4_1024_s4 unet_4x4_2048_s4_gate unet_8x8_1536 unet_16x16_768 unet_32x32_512 unet_32x32_res_conv unet_ln_ff2_2048"
for v in "" _r0 _r2 _nt _u8; do
  echo "== timing lib '$v'" >> $O/phases.log
  SF_TIMING_LIB=$L/libsf_fused_timing$v.so timeout 300 python tools/fconv_phases.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/phases.log
done
cat $O/phases.log
for cfg in "256 160 0" "1024 160 0" "512 160 1" "1024 160 1" "1024 320 1" "1024 640 1" "1024 100000 1" "1024 0 1"; do
  set -- $cfg
  echo "== scatter threads=$1 cutoff=$2 split=$3" >> $O/scatter.log
  SF_SC_THREADS=$1 SF_SC_CUTOFF=$2 SF_SC_SPLIT=$3 timeout 300 python tools/ngp_microbench.py 2>&1 | tail -2 >> $O/scatter.log
done
cat $O/scatter.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/unet_eval_b1_timeline.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/unet_eval_b1_kernel_stats.csv
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/ngp_microbench_kernel_stats.csv
grep "^#" $O/unet_eval_b1_timeline.txt | head -30
head -8 $O/ngp_microbench_kernel_stats.csv | cut -c1-140
