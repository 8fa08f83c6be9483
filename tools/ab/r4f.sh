cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4f}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -n 3
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "golden or bitwise or plms_canonical" 2>&1 | tail -n 3
for B in 1 4; do timeout 120 python tools/unet_time.py $B 2>&1 | grep "eval=" | tee -a $O/unet_time.log; done
timeout 300 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids | tee $O/graph_ablate_b1.log
timeout 300 python tools/graph_ablate.py 4 2>&1 | grep -v amdgpu.ids | tee $O/graph_ablate_b4.log
