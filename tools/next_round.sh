#!/bin/bash
# First GPU call of the next round: the measurements DESIGN.md section 8 asks for, each bounded by its own timeout.
#   gpurun --timeout 600 -- 'bash tools/next_round.sh r4a'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4a}; mkdir -p $O
export TMPDIR=/tmp
# 1. hand-off primitives: XCD-hierarchical grid barrier and software dependent launch vs kernel boundaries
for e in xcd_barrier pdl_chain; do
  timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/$e.hip -o /tmp/$e > $O/$e.build.log 2>&1 && timeout 60 /tmp/$e > $O/$e.log 2>&1
  echo "== $e"; tail -n 6 $O/$e.log
done
# 2. the opt-in kernels against the default path (VAE GroupNorm epilogue, composite backward, field forward on MFMA)
SF_TEST_EXPERIMENTAL=1 timeout 240 python -m pytest tests/test_gpu_experimental.py -q -m gpu > $O/experimental.log 2>&1; tail -n 3 $O/experimental.log
# 3. VAE timing with and without the GroupNorm epilogue
for v in 0 1; do echo "== SF_VAE_GN_EPI=$v"; SF_VAE_GN_EPI=$v timeout 90 python tools/vae_time.py 1 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee -a $O/vae_gn_epi.log; done
# 4. software dependent launch on the real UNet eval (variant library; each child process has its own timeout)
timeout 400 python tools/pdl_try.py > $O/pdl_try.log 2>&1; tail -n 5 $O/pdl_try.log
