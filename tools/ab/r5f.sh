cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
V=""
for o in 0 1 2; do for l in 0 1; do V="$V 0:SF_GN_SELF_EARLY_RING=$o:SF_GN_SELF_LEAN=$l"; done; done
timeout 600 python tools/fconv4_knockout.py 1 $V 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_order_lean_b1.log
timeout 600 python tools/fconv4_knockout.py 4 $V 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_order_lean_b4.log
