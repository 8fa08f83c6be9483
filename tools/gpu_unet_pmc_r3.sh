cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_unet}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/u1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/u2 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u3 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 6 > $O/u3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/unet_pmc_summary.py $O/r03_unet_eval_b1_pmc.json /tmp/u1 /tmp/u2 /tmp/u3 | head -30
