# Round-2 session-2 call 2: kernel-argument touch, epilogue operand hoist, GCA load reordering; nt / U=2 combinations.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2t}
mkdir -p $O
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/sparsefusion_amd
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_unet.py -q -x 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
for v in base noka nt ntu2 ntr0; do
  if [ $v = base ]; then lib=$L/libsparsefusion_hip.so; else lib=$L/libsparsefusion_hip_$v.so; fi
  echo "== variant $v" >> $O/unet_time.log
  SF_HIP_LIB=$lib timeout 300 python tools/unet_time.py 1 2>&1 | tail -2 >> $O/unet_time.log
done
cat $O/unet_time.log
CASES="unet_4x4_1024_s4 unet_4x4_2048_s4_gate unet_8x8_1536 unet_16x16_768 unet_32x32_512 unet_32x32_res_conv unet_ln_qkv_lazy"
for v in "" _nt _ntu2; do
  echo "== timing lib '$v'" >> $O/phases.log
  SF_TIMING_LIB=$L/libsf_fused_timing$v.so timeout 300 python tools/fconv_phases.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/phases.log
done
cat $O/phases.log
timeout 300 python tools/ngp_microbench.py 2>&1 | tail -2
cd /tmp
SF_HIP_LIB=$L/libsparsefusion_hip_nt.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/unet_eval_b1_timeline_nt.txt
cp $(find /tmp/rp1 -name "*kernel_stats.csv" | head -1) $O/unet_eval_b1_kernel_stats_nt.csv
grep "^#" $O/unet_eval_b1_timeline_nt.txt | head -32
