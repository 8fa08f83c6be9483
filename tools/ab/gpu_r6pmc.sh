# Round 6: counter passes over the eval loops at the closing commit (B = 1, 4, 32).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6pmc}
mkdir -p $O
export TMPDIR=/tmp
for B in 1 4 32; do
  N=6; [ $B = 32 ] && N=4
  (cd /tmp && timeout 900 python $GRAFT_REPO_ROOT/tools/unet_pmc.py $B $N $GRAFT_REPO_ROOT/$O/r06_unet_eval_b${B}_pmc.json > $GRAFT_REPO_ROOT/$O/pmc$B.log 2>&1)
  python -c "
import json; d=json.load(open('$O/r06_unet_eval_b${B}_pmc.json')); print($B, d['fused_conv_family'], d['whole_eval'])"
done
