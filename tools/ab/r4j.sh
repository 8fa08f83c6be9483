cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4j}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x -k "pipe" 2>&1 | tail -n 3
for B in 1 2 4; do
  echo "== B=$B all waves stage chunk 0" | tee -a $O/unet_time.log; timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
  echo "== B=$B staging waves only (r03)" | tee -a $O/unet_time.log
  SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_nochunk0.so timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
done
