# Round-6 opening set: parity of the plans the scaling lines run (B = 8 / 16 / 32, PLMS at B = 8), the specialised-kernel variants, and the
# in-graph ablation tables re-taken at HEAD.    bash tools/gpu_r6a.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6a}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch or batch8 or r05_specialised" > $O/r06_large_batch_parity.log 2>&1; tail -n 3 $O/r06_large_batch_parity.log
timeout 300 python tools/graph_ablate.py 1 > $O/r06_graph_ablate_b1_head.log 2>&1; head -n 3 $O/r06_graph_ablate_b1_head.log
timeout 300 python tools/graph_ablate.py 4 > $O/r06_graph_ablate_b4_head.log 2>&1; head -n 3 $O/r06_graph_ablate_b4_head.log
