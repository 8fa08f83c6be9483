#!/bin/bash
# r3 GPU call e: NGP chunked backward A/B, NGP + VAE + UNet parity tests on the new defaults, bench lines for configs 1 / 2 / 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
for c in 1 2 4 8; do echo "== SF_NGP_CHUNKS=$c" | tee -a $O/ngp.log; SF_NGP_CHUNKS=$c timeout 120 python tools/ngp_microbench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ngp.log; done
timeout 900 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_e2e_distill.py tests/test_gpu_occ_render.py tests/test_gpu_lpips.py tests/test_gpu_bench_multirank.py -m gpu -q > $O/tests.log 2>&1; tail -n 6 $O/tests.log
SF_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental.py -q -m gpu > $O/experimental.log 2>&1; tail -n 3 $O/experimental.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err; tail -c 1500 $O/bench_c1.json
timeout 300 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 1200 $O/bench_c3.json
timeout 300 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json
