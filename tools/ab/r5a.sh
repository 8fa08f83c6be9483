# r05 first GPU call: (1) cross-launch weight prefetch micro-benchmark, (2) the new parity tests, (3) the default bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/exp/weight_prefetch_chain.hip -o /tmp/wpc 2>$O/wpc_build.log && timeout 120 /tmp/wpc > $O/weight_prefetch_chain.log 2>&1
cat $O/weight_prefetch_chain.log
timeout 600 python -m pytest tests/test_gpu_ngp.py -x -q -s -k "bookkeeping or unequal" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/tests_ngp.log
timeout 300 python -m pytest tests/test_gpu_unet.py -x -q -k "time_table or medium or canonical" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests_unet.log
timeout 600 python bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r=json.load(open("gpurun_out/r5a/bench_n1.json"))
print("ms/step", r["ms_per_step"], "value", r["value"], "breakdown", r["breakdown_ms"])
print("roofline", {k:r["roofline"][k] for k in ("frac","frac_whole_eval","frac_whole_eval_survey_8d_bytes","avg_launch_us","fused_conv_ms_per_eval","traffic_over_algorithmic") if k in r["roofline"]})
print("also", json.dumps(r.get("also_measured"))[:1500])
PY
tail -5 $O/bench.err
