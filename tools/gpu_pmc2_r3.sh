# Counter passes for the NGP kernels, the VAE's LDS-tiled convs and the pipelined UNet convs -> $O/r03_ngp_vae_pmc.json
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc2}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE"
for t in "ngp tools/ngp_microbench.py" "vae tools/vae_time.py 1" "unet tools/unet_eval_loop.py 1 6"; do
  set -- $t; tag=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q_${tag}_f -- python $GRAFT_REPO_ROOT/$@ > $O/q_${tag}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q_${tag}_w -- python $GRAFT_REPO_ROOT/$@ > $O/q_${tag}_w.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/q_${tag}_s -- python $GRAFT_REPO_ROOT/$@ > $O/q_${tag}_s.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY > $O/r03_ngp_vae_pmc.json
import json, subprocess, sys
kern = {"ngp": ["k_ngp_field<0>", "k_ngp_field<1>", "k_ngp_field_bwd_mfma", "k_ngp_scatter<1024>", "k_ngp_scatter_fine", "k_ngp_composite_wave", "k_ngp_composite_bwd_wave"],
        "vae": ["k_conv_lds<8, false>", "k_conv_lds<4, false>", "k_conv_lds<8, true>", "k_conv_lds<4, true>", "k_conv_lds_gn<8, false>", "k_conv_lds_gn<4, false>", "k_gn_stats_px", "k_gn_apply", "k_gn_finalize"],
        "unet": ["k_conv_fused_pipe<2, 2, 12", "k_conv_fused_pipe<1, 1, 4", "k_conv_fused_pipe<1, 2, 6", "k_conv_fused<1, 1, 12, 1, 1", "k_gca_pool", "k_gca_net0", "k_gca_gate"]}
out = {"source": "tools/gpu_pmc2.sh: rocprofv3 --kernel-trace --pmc, three passes per target (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE); "
                 "means per dispatch.  FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE reads 1/2 of wide coalesced streams on gfx950 (MI355X_MICROARCH.md). "
                 "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the XCDs (GRBM / 8 / trace duration = 2.35 GHz on k_ngp_field_bwd_mfma, k_ngp_scatter)"}
for tag, names in kern.items():
    for n in names:
        j = json.loads(subprocess.check_output([sys.executable, "tools/pmc_collect.py", f"/tmp/q_{tag}_f", n, f"/tmp/q_{tag}_w", f"/tmp/q_{tag}_s"]))
        d = {k: v["mean_per_dispatch"] for k, v in j.items()}
        d["dispatches"] = max([v["dispatches"] for v in j.values()] or [0])
        if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
            d["valu_issue_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"]
        out[n] = d
print(json.dumps(out, indent=1))
PY
python -c "
import json; d=json.load(open('$O/r03_ngp_vae_pmc.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, {a: (round(b,4) if isinstance(b,float) and b<10 else round(b)) for a,b in v.items() if a in ('FETCH_SIZE','WRITE_SIZE','mfma_busy_frac','wait_any_frac','valu_issue_frac','dispatches')})"
