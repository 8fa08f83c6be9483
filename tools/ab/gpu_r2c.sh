cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2c}
mkdir -p $O
python tools/fconv_phases.py > $O/phases.log 2>&1
cat $O/phases.log
