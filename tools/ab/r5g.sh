cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fconv4_knockout.py 4 0:SF_FCONV_MINW=2 0:SF_FCONV_MINW=4 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_minw_b4.log
timeout 600 python tools/fconv4_knockout.py 1 0:SF_FCONV_MINW=2 0:SF_FCONV_MINW=4 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_minw_b1.log
timeout 600 python tools/fconv4_knockout.py 2 0:SF_FCONV_MINW=2 0:SF_FCONV_MINW=4 2>&1 | grep -v amdgpu.ids | tee $O/fconv4_minw_b2.log
