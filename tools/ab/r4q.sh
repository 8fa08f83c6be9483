# r04 call 25: checkpoint after the binned NGP scatter: NGP tests (incl. the full-size binned-vs-atomics test), default bench line,
# NGP kernel trace + counters of the new kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4q}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_occ_render.py -m gpu -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
timeout 400 python bench.py > $O/r04_bench_n1.json 2> $O/bench_n1.err; tail -n 1 $O/r04_bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'), d['roofline'].get('frac'), d['also_measured']['config3_B4']['ms_per_step'] if 'also_measured' in d else None)"
cd /tmp
O=$GRAFT_REPO_ROOT/$O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r04_ngp_microbench_kernel_stats.csv; tail -n 3 $O/rpn.log
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/n1 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/n2 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/n3 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n3.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 k_ngp /tmp/n2 /tmp/n3 > $O/r04_ngp_pmc_all_kernels.json
for k in k_ngp_field_bwd_mfma k_ngp_bin_reduce "k_ngp_bin(" "k_ngp_scatter<" "k_ngp_field<"; do echo "== $k"; python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 "$k" /tmp/n2 /tmp/n3 | python -c "import sys,json; d=json.load(sys.stdin); print({k: round(v['mean_per_dispatch']) for k,v in d.items()})"; done | tee $O/r04_ngp_pmc_by_kernel.log
