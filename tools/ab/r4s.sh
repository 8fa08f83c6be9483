# r04 call 32: hybrid-plan threshold (Unet.unfused_min_rows) at B = 2 / 4 / 8; fused GPU cases at HEAD
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4s}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "gca" 2>&1 | tail -n 3
{
for spec in "4 8192" "4 4096" "4 2048" "2 8192" "2 2048" "8 8192" "8 2048"; do set -- $spec; echo "== B=$1 unfused_min_rows=$2"; SF_UNET_ATTRS=unfused_min_rows=$2 timeout 100 python tools/unet_time.py $1 2>&1 | grep "sampler path"; done
} | tee $O/unet_hybrid_rows_ab.log
