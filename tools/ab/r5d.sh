cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fconv4_knockout.py 1 0 8 0w4 8w4 1w4 3w4 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_knockout_waves_b1.log
timeout 600 python -m pytest tests/test_gpu_ngp.py -q -s -k "bookkeeping" 2>&1 | grep "bookkeeping \|passed\|failed" | tee $O/tests_ngp.log
