cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2p}
mkdir -p $O
python -m pytest tests/test_gpu_occ_render.py tests/test_gpu_occ.py tests/test_gpu_bench_multirank.py tests/test_gpu_lpips.py tests/test_gpu_ngp.py -q 2>&1 | tail -12 > $O/t1.log
python -m pytest tests/test_gpu_unet.py -q -s -k "fast_path" 2>&1 | grep "B=\|passed\|failed" > $O/t2.log
python tools/occ_eval_time.py > $O/occ_eval.log 2>&1
python tools/unet_time.py 1 > $O/unet_time1.log 2>&1
cat $O/t1.log $O/t2.log; tail -n 2 $O/occ_eval.log $O/unet_time1.log
