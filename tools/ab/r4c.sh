cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c}; mkdir -p $O
timeout 400 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids | tee $O/graph_ablate_b1.log
