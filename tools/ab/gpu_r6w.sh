# Round 6: LPIPS on the IEEE-half operand build (gradient error vs the fp32 oracle over gradient scales).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6w}
mkdir -p $O
timeout 600 python tools/exp/lpips_operand_probe.py 2>&1 | grep -v Warning > $O/r06_lpips_operand_probe.log; cat $O/r06_lpips_operand_probe.log
