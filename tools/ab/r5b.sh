cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fconv4_knockout.py 1 0 1 2 3 4 5 6 7 8 9 1062 1766 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_knockout_b1.log
timeout 600 python -m pytest tests/test_gpu_ngp.py -x -q -s -k "bookkeeping or unequal" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/tests_ngp.log
