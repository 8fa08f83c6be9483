# Round-3 closing measurement set (second half of the round: LDS-DMA convs): full GPU suite, bench lines, kernel trace of the step,
# counter passes for the new conv kernels.  bash tools/gpu_final_r3b.sh <tag>; results in gpurun_out/<tag>/, copied to profiles/ afterwards.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final3c}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; tail -n 2 $O/r03_gpu_tests.log
SF_OPERAND=f16 timeout 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q > $O/r03_gpu_tests_f16_process.log 2>&1; tail -n 2 $O/r03_gpu_tests_f16_process.log
timeout 300 python bench.py > $O/r03_bench_n1.json 2> $O/bench_n1.err
timeout 200 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_views4.json 2> $O/bench_v4.err
timeout 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_n1_config2.json 2> $O/bench_c2.err
timeout 200 python bench.py --total-views 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_n1_total32.json 2> $O/bench_t32.err
for f in r03_bench_n1 r03_bench_n1_views4 r03_bench_n1_config2 r03_bench_n1_total32; do tail -n 1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', {k:d[k] for k in ('value','ms_per_step','scaling')}, d.get('breakdown_ms'), d.get('roofline_mfma'))"; done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
cp $(find /tmp/rp2 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r03_bench_kernel_stats.csv
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q_vae_f -- python $GRAFT_REPO_ROOT/tools/vae_time.py 1 > $GRAFT_REPO_ROOT/$O/q_vae_f.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q_vae_w -- python $GRAFT_REPO_ROOT/tools/vae_time.py 1 > $GRAFT_REPO_ROOT/$O/q_vae_w.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/q_vae_s -- python $GRAFT_REPO_ROOT/tools/vae_time.py 1 > $GRAFT_REPO_ROOT/$O/q_vae_s.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY > $O/r03_vae_halo_pmc.json
import json, subprocess, sys
names = ["k_conv3_halo<8, 4, true>", "k_conv3_halo<8, 4, false>", "k_conv3_halo<4, 4, true>", "k_conv3_halo<4, 4, false>", "k_conv_glds<8, 4, false>", "k_conv_glds<4, 4, false>", "k_conv_glds<8, 4, true>",
         "k_conv_lds<8, true>", "k_conv_lds<4, true>", "k_conv_lds_gn<8, true>", "k_gn_apply", "k_gn_finalize", "k_conv_igemm"]
out = {"source": "tools/gpu_final_r3b.sh: rocprofv3 --kernel-trace --pmc on tools/vae_time.py 1, three passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE); means per dispatch. "
                 "FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE reads 1/2 of wide coalesced streams on gfx950, MI355X_MICROARCH.md). "
                 "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"}
for n in names:
    j = json.loads(subprocess.check_output([sys.executable, "tools/pmc_collect.py", "/tmp/q_vae_f", n, "/tmp/q_vae_w", "/tmp/q_vae_s"]))
    d = {k: v["mean_per_dispatch"] for k, v in j.items()}
    if not d:
        continue
    d["dispatches"] = max([v["dispatches"] for v in j.values()] or [0])
    if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    out[n] = d
print(json.dumps(out, indent=1))
PY
python -c "
import json; d=json.load(open('$O/r03_vae_halo_pmc.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, {a: (round(b,4) if isinstance(b,float) and b<10 else round(b)) for a,b in v.items() if a in ('FETCH_SIZE','WRITE_SIZE','mfma_busy_frac','wait_any_frac','SQ_LDS_BANK_CONFLICT','dispatches')})"
python tools/unet_time.py 1 2>&1 | grep "sampler" > $O/unet_time1.log; python tools/unet_time.py 4 2>&1 | grep sampler > $O/unet_time4.log; python tools/vae_time.py 1 2>&1 | grep "^B=" > $O/vae_time.log
cat $O/unet_time1.log $O/unet_time4.log $O/vae_time.log
