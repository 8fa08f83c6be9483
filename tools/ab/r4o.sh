# r04 call 17: (a) res_conv beside the GlobalContext pooling launch (4x4 level) on / off; (b) binned NGP table-gradient scatter
# against the atomics path, cut-off sweep; parity tests of both first
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4o}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_unet.py -m gpu -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
{
echo "== atomics (SF_NGP_BIN=0)"; SF_NGP_BIN=0 timeout 100 python tools/ngp_microbench.py
for c in 60 112; do echo "== binned, cut-off $c"; SF_NGP_BIN_CUTOFF=$c timeout 100 python tools/ngp_microbench.py; done
} 2>&1 | tee $O/ngp_bin_ab.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $GRAFT_REPO_ROOT/$O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/ngp_bin_kernel_stats.csv; head -8 $GRAFT_REPO_ROOT/$O/ngp_bin_kernel_stats.csv | cut -c1-150
