# Round 6: k_conv3_halo_sm (whole 4x4 / 8x8 maps): op cases, plan parity, timings incl. the 8x8 level off the fused kernels.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6k}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "split_k or pixel_shuffle or lds or glds or halo" > $O/ops.log 2>&1; tail -n 5 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 20
for attrs in "" "unfused_min_rows_4=128" "unfused_min_rows_8=512" "unfused_min_rows_8=2048"; do
  for B in 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_halo_sm_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_halo_sm_ab.log
  done
done
cat $O/r06_halo_sm_ab.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 16 $O/r06_graph_ablate_b32.log
SF_UNET_ATTRS=unfused_min_rows_8=2048 timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32_u8.log; head -n 16 $O/r06_graph_ablate_b32_u8.log
