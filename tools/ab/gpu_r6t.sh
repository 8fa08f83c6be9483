# Round 6: the thresholds of the small levels once more, with k_gn_one on small tensors and k_gemm_rows_ks in place (B = 4 / 8).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6t}
mkdir -p $O
export TMPDIR=/tmp
for attrs in "" "unfused_min_rows_4=128" "unfused_min_rows_4=128,unfused_lin_min_rows=128" "unfused_min_rows_8=512" "unfused_min_rows_4=128,unfused_min_rows_8=512,unfused_lin_min_rows=128"; do
  echo "== SF_UNET_ATTRS=$attrs B=8" >> $O/r06_small_levels_ab.log
  SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py 8 2>&1 | grep "sampler path" >> $O/r06_small_levels_ab.log
done
for attrs in "" "lds_mid_min_batch=4,lds_mid_min_rows=64,unfused_min_rows_4=64" "lds_mid_min_batch=4,lds_mid_min_rows=64,unfused_min_rows_4=64,unfused_lin_min_rows=64"; do
  echo "== SF_UNET_ATTRS=$attrs B=4" >> $O/r06_small_levels_ab.log
  SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py 4 2>&1 | grep "sampler path\|Error\|error" >> $O/r06_small_levels_ab.log
done
cat $O/r06_small_levels_ab.log
