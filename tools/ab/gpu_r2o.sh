cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2o}
mkdir -p $O
python -m pytest tests/test_gpu_e2e_distill.py -q -s 2>&1 | tail -15 > $O/t_e2e.log
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/t_all.log
cat $O/t_e2e.log; tail -n 8 $O/t_all.log
