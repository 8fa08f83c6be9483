#!/bin/bash
# r3 GPU call cc: what the feature re-gather costs in k_ngp_field_bwd_mfma (measurement builds; serial backward so that kernel times add up)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3cc; mkdir -p $O
for v in prod fb_x1 fb_x2; do
  L=""; [ $v != prod ] && L=$PWD/sparsefusion_amd/libsparsefusion_hip_$v.so
  echo "== $v" | tee -a $O/ngp.log
  SF_HIP_LIB=$L SF_NGP_OVERLAP=0 timeout 120 python tools/ngp_microbench.py 2>&1 | grep render | tee -a $O/ngp.log
done
