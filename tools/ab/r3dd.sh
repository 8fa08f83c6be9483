#!/bin/bash
# r3 GPU call dd: NGP field cache (forward keeps the features, backward skips the re-gather): parity + render time, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3dd; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_occ_render.py -m gpu -q > $O/tests_ngp.log 2>&1; tail -n 3 $O/tests_ngp.log
timeout 300 python -m pytest tests/test_gpu_e2e_distill.py -m gpu -q -k small > $O/tests_e2e.log 2>&1; tail -n 2 $O/tests_e2e.log
for c in 1 0; do echo "== SF_NGP_FEAT_CACHE=$c" | tee -a $O/ngp.log; SF_NGP_FEAT_CACHE=$c timeout 120 python tools/ngp_microbench.py 2>&1 | grep render | tee -a $O/ngp.log; done
