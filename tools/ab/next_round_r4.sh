# First GPU call of the next round (about 8 GPU-minutes): the measurements round 3 ran out of budget for.
#   bash tools/next_round_r4.sh <tag>      results in gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4a}; mkdir -p $O
export TMPDIR=/tmp
# 1. NGP after the field cache: per-kernel times of render fwd + bwd (what bounds the 3.75 ms backward now?)
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r04_ngp_microbench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
head -8 $O/r04_ngp_microbench_kernel_stats.csv | cut -c1-100
# 2. the 64-tile halo layers (SD-VAE 32 x 32, 18 of them at 0.07 of the MFMA peak): one layer alone, every kernel
timeout 120 python tools/conv_time.py 1 4 6 7 --shape=4 2>&1 | grep -v amdgpu.ids | tee $O/conv_time_32x32.log
# 3. UNet hybrid threshold (GroupNorm + k_conv3_halo for large-M ResnetBlocks): B = 4 / 16 around the default of 8192 rows
for B in 4 16; do for thr in 1073741824 2048 4096 8192; do
  echo -n "B=$B SF_UNFUSED_ROWS=$thr: " | tee -a $O/hybrid_sweep.log; SF_UNFUSED_ROWS=$thr timeout 120 python tools/unet_time.py $B 2>&1 | grep "sampler path" | tee -a $O/hybrid_sweep.log
done; done
# 4. where the B = 32 eval (7.8 ms) goes after the hybrid plan
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpu -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 32 10 > $O/rpu.log 2>&1
cp $(find /tmp/rpu -name "*kernel_stats.csv" | head -1) $O/r04_unet_eval_b32_kernel_stats.csv
cd $GRAFT_REPO_ROOT
head -12 $O/r04_unet_eval_b32_kernel_stats.csv | cut -c1-110
# 5. what k_ngp_field_bwd_mfma (2.0 ms per render, 22 % MFMA busy) waits for: LDS counters (the 16-points-per-trip variant was slower,
#    which points at the 4-byte fragment reads rather than at latency cover)
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/q_ngp_l -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/q_ngp_l.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_collect.py /tmp/q_ngp_l k_ngp_field_bwd_mfma > $O/r04_ngp_field_bwd_lds_pmc.json; python -c "
import json; d=json.load(open('$O/r04_ngp_field_bwd_lds_pmc.json')); print({k: round(v['mean_per_dispatch']) for k, v in d.items()})"
