# Round 6: k_conv3s (csrc/fused_conv3s.h) -- parity of its op cases and of the whole eval, then the A/B in the replayed graph:
#   default (2-D tiles) | full-width strips | the general pipelined kernel.    bash tools/gpu_r6b.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6b}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -s -k "conv3s" > $O/conv3s_cases.log 2>&1; tail -n 3 $O/conv3s_cases.log
timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "forward_matches_reference_golden or bitwise" > $O/unet_golden.log 2>&1; tail -n 3 $O/unet_golden.log
for attrs in "" "conv3s_tw32=0,conv3s_tw16=0" "conv3s_tw32=0" "conv3s_tw16=0" "conv3s=0"; do
  for B in 1 2 4; do
    echo "== SF_UNET_ATTRS=$attrs B=$B" >> $O/r06_conv3s_ab.log
    SF_UNET_ATTRS=$attrs timeout 200 python tools/unet_time.py $B 2>&1 | grep -v amdgpu >> $O/r06_conv3s_ab.log
  done
done
cat $O/r06_conv3s_ab.log
timeout 300 python tools/graph_ablate.py 1 > $O/r06_graph_ablate_b1_conv3s.log 2>&1; head -n 30 $O/r06_graph_ablate_b1_conv3s.log
