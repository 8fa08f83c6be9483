cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4l}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused.py -q -x -k "pair" 2>&1 | tail -n 3
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "golden or bitwise or lazy" 2>&1 | tail -n 3
for B in 1 2 4; do
  echo "== B=$B res_conv merged into conv1's workgroups" | tee -a $O/unet_time.log; timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
  echo "== B=$B two sets of workgroups (r03)" | tee -a $O/unet_time.log
  SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_norc.so timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log
done
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep "pair\|full" | tee $O/ablate_pairs.log
