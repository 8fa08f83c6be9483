cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4g}; mkdir -p $O
for B in 1 4; do timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler" | tee -a $O/unet_time.log; done
timeout 600 python tools/tile_sweep.py 4 2>&1 | grep -v amdgpu.ids | tee $O/tile_sweep_b4.log
