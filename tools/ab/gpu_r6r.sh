# Round 6: the ragged-batch parity test; counter passes (FETCH / WRITE / SQ) over the B = 32 and B = 1 eval loops at HEAD.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6r}
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "ragged" > $O/ragged.log 2>&1; grep "worst image\|passed\|failed\|Error" $O/ragged.log | tail -n 10
cd /tmp && timeout 900 python $GRAFT_REPO_ROOT/tools/unet_pmc.py 32 4 $GRAFT_REPO_ROOT/$O/r06_unet_eval_b32_pmc.json > $GRAFT_REPO_ROOT/$O/pmc32.log 2>&1; tail -n 30 $GRAFT_REPO_ROOT/$O/pmc32.log
cd /tmp && timeout 600 python $GRAFT_REPO_ROOT/tools/unet_pmc.py 1 6 $GRAFT_REPO_ROOT/$O/r06_unet_eval_b1_pmc.json > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1; tail -n 5 $GRAFT_REPO_ROOT/$O/pmc1.log
