# First GPU call of the next round (about 4 GPU-minutes): the measurements round 4 ended on.
#   bash tools/next_round_r5.sh <tag>      results in gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5a}; mkdir -p $O
export TMPDIR=/tmp
# 1. the micro-benchmarks that found this round's two largest effects (LDS atomic rates by type, 16-byte store patterns): re-taken on the
#    box of the day, as the baseline for (a) wave-contiguous sub-runs in k_ngp_bin, (b) any LDS accumulator elsewhere
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/exp/lds_atomic_rate.hip -o /tmp/lds_atomic_rate 2>/dev/null && timeout 60 /tmp/lds_atomic_rate | tee $O/lds_atomic_rate.log
hipcc --offload-arch=gfx950 -O3 tools/exp/store_patterns.hip -o /tmp/store_patterns 2>/dev/null && timeout 60 /tmp/store_patterns | tee $O/store_patterns.log
# 2. NGP render: per-kernel times (field backward 0.66 / bin 0.59 / reduce 0.24 per 8192-ray chunk, cached-level scatter 0.60 at the end of r04)
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r05_ngp_microbench_kernel_stats.csv; grep render $O/rpn.log
# 3. LDS counters of the fused UNet convs again (r04: LDS active 5-8 % of a launch, profiles/r04_unet_lds_pmc_by_kernel.log), one pass, kernel trace only
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u4 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u4.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/u4 k_conv_fused > $O/r05_unet_fconv_lds_pmc.json; head -c 600 $O/r05_unet_fconv_lds_pmc.json
# 4. the in-graph cost table the round starts from
cd $GRAFT_REPO_ROOT
timeout 200 python tools/graph_ablate.py 1 2>&1 | grep -v amdgpu.ids > $O/r05_graph_ablate_b1.log; head -8 $O/r05_graph_ablate_b1.log
