cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2q}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ngp.py tests/test_gpu_e2e_distill.py -q -s 2>&1 | grep -v "^$" | tail -12 > $O/t_ngp.log
python tools/ngp_microbench.py > $O/mb_mfma.log 2>&1
SF_NGP_BWD_VALU=1 python tools/ngp_microbench.py > $O/mb_valu.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r02_ngp_microbench_kernel_stats.csv
cat $O/t_ngp.log; tail -n 3 $O/mb_mfma.log $O/mb_valu.log
head -12 $O/r02_ngp_microbench_kernel_stats.csv | cut -c1-150
