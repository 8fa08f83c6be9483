# Round 6: the e2e distillation parity test with LPIPS on bf16 operands (r05 arithmetic), for comparison with the IEEE-half default.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6y}
mkdir -p $O
timeout 900 python tools/exp/e2e_with_bf16_lpips.py > $O/e2e_bf16.log 2>&1; grep "after\|passed\|failed" $O/e2e_bf16.log
