cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
for k in 2 4 8; do timeout 300 python tools/exp/multistream_evals.py $k 30 2>&1 | grep "K = "; done | tee $O/multistream_evals.log
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/exp/multistream_evals.py 8 30 2>&1 | grep "K = " | sed 's/^/GPU_MAX_HW_QUEUES=8  /' | tee -a $O/multistream_evals.log
