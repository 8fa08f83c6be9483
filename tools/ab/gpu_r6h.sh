# Round 6: the f16 build at HEAD, ablation tables at B = 8 / 32 (where the large-batch plans spend their time).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6h}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "fp16" > $O/f16.log 2>&1; tail -n 3 $O/f16.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 30 $O/r06_graph_ablate_b32.log
timeout 300 python tools/graph_ablate.py 8 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b8.log; head -n 30 $O/r06_graph_ablate_b8.log
