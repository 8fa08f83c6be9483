# Round 4 profile set: kernel traces + counter passes of the B = 1 UNet eval, the NGP render, the EFT feature render.   bash tools/gpu_profiles_r4.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof4}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpu -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 50 > $O/rpu.log 2>&1
cp $(find /tmp/rpu -name "*kernel_stats.csv" | head -1) $O/r04_unet_eval_b1_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/rpu $O/r04_unet_eval_b1_timeline.txt; tail -n 32 $O/r04_unet_eval_b1_timeline.txt | head -n 4
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/u1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/u2 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u3 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u3.log 2>&1
python $GRAFT_REPO_ROOT/tools/unet_pmc_summary.py $O/r04_unet_eval_b1_pmc.json /tmp/u1 /tmp/u2 /tmp/u3 | head -n 16
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpn -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/rpn.log 2>&1
cp $(find /tmp/rpn -name "*kernel_stats.csv" | head -1) $O/r04_ngp_microbench_kernel_stats.csv; tail -n 3 $O/rpn.log
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/n1 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/n2 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/n3 -- python $GRAFT_REPO_ROOT/tools/ngp_microbench.py > $O/n3.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 k_ngp /tmp/n2 /tmp/n3 > $O/r04_ngp_pmc_all_kernels.json
for k in k_ngp_field_bwd_mfma k_ngp_scatter_fine "k_ngp_scatter<" "k_ngp_field<"; do echo "== $k"; python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/n1 "$k" /tmp/n2 /tmp/n3 | python -c "import sys,json; d=json.load(sys.stdin); print({k: round(v['mean_per_dispatch']) for k,v in d.items()})"; done | tee $O/r04_ngp_pmc_by_kernel.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpe -- python $GRAFT_REPO_ROOT/tools/eft_time.py 6 > $O/rpe.log 2>&1
cp $(find /tmp/rpe -name "*kernel_stats.csv" | head -1) $O/r04_eft_render_kernel_stats.csv; tail -n 4 $O/rpe.log; head -n 12 $O/r04_eft_render_kernel_stats.csv | cut -c1-120
