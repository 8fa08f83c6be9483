# Round-4 closing measurement set (after the binned NGP scatter): tools/gpu_final_r4.sh + the NGP kernel trace and counters.
#   bash tools/gpu_final_r4b.sh <tag>      results in gpurun_out/<tag>/ (copied to profiles/ as r04_*)
bash $GRAFT_REPO_ROOT/tools/gpu_final_r4.sh ${1:-final4c}
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-final4c}
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r04_ngp_microbench_kernel_stats.csv; grep "render" $O/rpn.log
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/n1 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/n2 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/n3 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n3.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 k_ngp /tmp/n2 /tmp/n3 > $O/r04_ngp_pmc_all_kernels.json
for k in k_ngp_field_bwd_mfma k_ngp_bin_reduce "k_ngp_bin(" "k_ngp_scatter<" "k_ngp_field<"; do echo "== $k"; python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 "$k" /tmp/n2 /tmp/n3 | python -c "import sys,json; d=json.load(sys.stdin); print({k: round(v['mean_per_dispatch']) for k,v in d.items()})"; done > $O/r04_ngp_pmc_by_kernel.log
