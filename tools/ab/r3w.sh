#!/bin/bash
# r3 GPU call w: cache counters of k_conv_glds on the 128x128 256->256 layer: product vs the A-hot measurement build (x6)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCC\|TCP\|TA\|TD\|SQ_LDS\|SQ_INSTS_LDS\|SQ_WAIT\)_[A-Za-z0-9_]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
G1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G2="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
G3="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
for v in prod x6; do
  if [ $v = x6 ]; then export SF_HIP_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip_glds_x6.so; fi
  n=0
  for g in "$G1" "$G2" "$G3" "$G4"; do n=$((n+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/w_${v}_$n -- python $GRAFT_REPO_ROOT/tools/conv_time.py 4 --shape=0 > $O/w_${v}_$n.log 2>&1 || echo "pass $v $n failed: $(tail -n 2 $O/w_${v}_$n.log)"
  done
  python $GRAFT_REPO_ROOT/tools/pmc_collect.py /tmp/w_${v}_1 k_conv_glds /tmp/w_${v}_2 /tmp/w_${v}_3 /tmp/w_${v}_4 > $O/pmc_$v.json
  python -c "
import json; d=json.load(open('$O/pmc_$v.json')); print('$v', {k: round(x['mean_per_dispatch']) for k,x in d.items()})"
done
