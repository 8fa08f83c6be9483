#!/bin/bash
# wave-per-ray composite + many-row time table GEMM: parity, then the bench with whatever passed
O=gpurun_out/r3l; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_unet_ops.py -q -m gpu -k "gemv" > $O/t_gemv.log 2>&1; G=$?; tail -n 2 $O/t_gemv.log
timeout 200 python -m pytest tests/test_gpu_ngp.py tests/test_gpu_e2e_distill.py -q -m gpu > $O/t_ngp.log 2>&1; C=$?; tail -n 2 $O/t_ngp.log
[ $G -ne 0 ] && export SF_GEMM_ROWS=0
[ $C -ne 0 ] && export SF_COMPOSITE_WAVE=0
echo "knobs: SF_GEMM_ROWS=${SF_GEMM_ROWS:-1} SF_COMPOSITE_WAVE=${SF_COMPOSITE_WAVE:-1}" | tee $O/knobs.txt
timeout 200 python -m pytest tests/test_gpu_unet.py -q -m gpu -k "sampler_fast or plms" > $O/t_unet.log 2>&1; tail -n 2 $O/t_unet.log
timeout 60 python tools/ngp_microbench.py 2>&1 | grep render | tee $O/ngp_mb.log
timeout 200 python bench.py > $O/bench_n1.json 2> $O/bench.err
tail -n 1 $O/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('breakdown_ms'))"
