cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fconv4_knockout.py 1 5 6 7 8 9 1062 1766 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_knockout_b1.log
timeout 900 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_occ_render.py tests/test_gpu_e2e_distill.py -q -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/tests_ngp.log
