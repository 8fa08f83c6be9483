cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3f}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
bash tools/gpu_final.sh ${1:-r3f}
