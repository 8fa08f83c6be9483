# Round 6: split-K / pixel shuffle on the LDS-tiled kernels (the 4x4 level and the Upsample convs of the B >= 8 plans): op cases, plan parity, timings.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6i}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "split_k or pixel_shuffle or lds or glds" > $O/ops.log 2>&1; tail -n 5 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch or plms_batch8" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 20
for attrs in "" "lds_mid_min_rows=0" "unfused_min_rows_4=0" "unfused_min_rows_8=512" "lds_mid_min_batch=4"; do
  for B in 4 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_mid_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_mid_ab.log
  done
done
cat $O/r06_mid_ab.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 24 $O/r06_graph_ablate_b32.log
timeout 400 python tools/graph_ablate.py 8 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b8.log; head -n 24 $O/r06_graph_ablate_b8.log
