# Round 6: 8-row GEMVs (the GlobalContext MLPs of a B = 8 hybrid block) on k_gemm_rows_ks: op cases, B = 8 parity (UNet + 51-eval PLMS), eval time.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6u}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_unet_ops.py -m gpu -q -k "gemv" > $O/ops.log 2>&1; tail -n 3 $O/ops.log
for k in 1 2; do timeout 200 python tools/unet_time.py 8 2>&1 | grep "sampler path" >> $O/r06_unet_time_b8.log; done; cat $O/r06_unet_time_b8.log
timeout 1200 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "large_batch or plms_batch8 or ragged" > $O/b8.log 2>&1; grep "rel L2\|worst image\|passed\|failed\|Error" $O/b8.log | tail -n 24
timeout 300 python tools/graph_ablate.py 8 2>&1 | grep -v amdgpu > $O/r06_graph_ablate_b8.log; head -n 16 $O/r06_graph_ablate_b8.log
