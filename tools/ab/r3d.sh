#!/bin/bash
# r3 GPU call d: full GPU suite on the cleaned-up library, NGP backward overlap A/B, VAE GroupNorm-epilogue A/B, opt-in kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
for v in 1 0; do echo "== SF_NGP_OVERLAP=$v" | tee -a $O/ngp.log; SF_NGP_OVERLAP=$v timeout 120 python tools/ngp_microbench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ngp.log; done
for v in 0 1; do echo "== SF_VAE_GN_EPI=$v" | tee -a $O/vae.log; SF_VAE_GN_EPI=$v timeout 120 python tools/vae_time.py 1 2>&1 | grep "^B=" | tee -a $O/vae.log; done
SF_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental.py -q -m gpu > $O/experimental.log 2>&1; tail -n 3 $O/experimental.log
