cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fconv4_knockout.py 1 0:SF_GN_SELF_EARLY_RING=0 0:SF_GN_SELF_EARLY_RING=1 1:SF_GN_SELF_EARLY_RING=1 3:SF_GN_SELF_EARLY_RING=1 4:SF_GN_SELF_EARLY_RING=1 8 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_early_ring_b1.log
timeout 600 python tools/fconv4_knockout.py 4 0:SF_GN_SELF_EARLY_RING=0 0:SF_GN_SELF_EARLY_RING=1 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_early_ring_b4.log
