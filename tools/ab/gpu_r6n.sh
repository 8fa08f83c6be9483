# Round 6: the large-batch parity cases at HEAD (unfused 4x4 linears from B = 16), per-level thresholds of the hybrid ResnetBlocks.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6n}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch or plms_batch8 or specialised" > $O/large_batch.log 2>&1; grep "rel L2\|passed\|failed\|Error" $O/large_batch.log | tail -n 30
for attrs in "" "unfused_min_rows_16=4096" "unfused_min_rows_16=2048" "unfused_min_rows_32=4096"; do
  for B in 4 8 16 32; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_levels_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_levels_ab.log
  done
done
cat $O/r06_levels_ab.log
