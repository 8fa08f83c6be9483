# Round 6: weight-major XCD tile order in k_conv3_halo_sm: op cases, eval times B = 16 / 32 (two runs each), ablation at B = 32, FETCH counter of the class.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6s}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "split_k" > $O/ops.log 2>&1; tail -n 3 $O/ops.log
for k in 1 2; do for B in 8 16 32; do timeout 200 python tools/unet_time.py $B 2>&1 | grep "sampler path" >> $O/r06_unet_time.log; done; done
cat $O/r06_unet_time.log
timeout 400 python tools/graph_ablate.py 32 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b32.log; head -n 12 $O/r06_graph_ablate_b32.log
timeout 400 python tools/graph_ablate.py 16 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b16.log; head -n 10 $O/r06_graph_ablate_b16.log
cd /tmp && timeout 900 python $GRAFT_REPO_ROOT/tools/unet_pmc.py 32 4 $GRAFT_REPO_ROOT/$O/r06_unet_eval_b32_pmc.json > $GRAFT_REPO_ROOT/$O/pmc32.log 2>&1
cd $GRAFT_REPO_ROOT; python -c "
import json; d=json.load(open('$O/r06_unet_eval_b32_pmc.json'))['per_class']
for k in ('k_conv3_halo_sm','k_conv3_halo','k_conv_glds'): print(k, d[k])"
