"""Counter passes over the UNet eval (r06; north_star: "rocprof HBM GB/s and MFMA-busy counters reported against gfx950 peak").

Three rocprofv3 runs of `tools/unet_eval_loop.py B N` (plain launches of the sampler's eval body), each `--kernel-trace --pmc <one group>` and
nothing else (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; counters in their own run):
    FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
and a summary per kernel class: dispatches per eval, mean duration (the kernel trace of the same runs), HBM bytes fetched (KiB x 1024 x 2: on
gfx950 FETCH_SIZE reports half of a wide coalesced streaming read) and written (uncalibrated), GB/s from those counters, MFMA-busy fraction
= SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), wave-parked share SQ_WAIT_ANY / SQ_WAVE_CYCLES.

usage: unet_pmc.py [B] [evals] [out.json]          (also imported by bench.py: collect() / summarise())"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
FUSED = ("k_conv_fused", "k_gca_pool_rc", "k_conv4_gn", "k_lin4_ln", "k_lin4_attn", "k_conv3s")      # the fused [norm +] conv launches: bench.py's `roofline` family


def kclass(name):
    n = name.replace("void ", "").split("(")[0]
    for key in ("k_conv3s", "k_conv_fused_pipe_rc", "k_conv_fused_pipe_pair", "k_conv_fused_pipe", "k_conv_fused_pair", "k_conv4_gn_mb", "k_conv4_gn", "k_lin4_ln", "k_lin4_attn",
                "k_gca_pool_rc", "k_conv_fused", "k_gca_net0", "k_gca_gate", "k_gca_pool", "k_gca_logits", "k_conv_igemm", "k_conv_lds", "k_conv_glds",
                "k_conv3_halo_sm", "k_conv3_halo", "k_gemm_rows_ks", "k_gemm_rows", "k_gemv", "k_attn16", "k_layernorm",
                "k_splitk_reduce", "k_slots", "k_init_x", "k_gn_one", "k_gn_"):      # (r06: the kernels of the large-batch plans)
        if key in n:
            return key
    return "other"


def collect(B=1, evals=6, timeout_s=300, groups=GROUPS, target=None):
    """Run the passes; returns {kernel name: {"n": dispatches, "dur_ns": sum, counter: sum, ...}} or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    target = target or [sys.executable, os.path.join(ROOT, "tools", "unet_eval_loop.py"), str(B), str(evals)]
    acc = defaultdict(lambda: defaultdict(float))
    for gi, grp in enumerate(groups):
        out = tempfile.mkdtemp(prefix="sf_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", *grp, "--output-format", "csv", "-d", out, "--"] + target, cwd="/tmp", env=env,
                           timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    a = acc[r["Kernel_Name"]]
                    a[r["Counter_Name"]] += float(r["Counter_Value"])
                    a["n_" + r["Counter_Name"]] += 1
            if gi == 0:                                        # durations and dispatch counts from the first pass's kernel trace
                for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        a = acc[r["Kernel_Name"]]
                        a["n"] += 1
                        a["dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return acc


def summarise(acc, evals, pred=None):
    """Per-class (or, with `pred`, one aggregate over the kernels pred(name) selects) per-dispatch means."""
    cls = defaultdict(lambda: defaultdict(float))
    for name, a in acc.items():
        if "rocclr" in name or not a.get("n"):
            continue
        key = "selected" if pred else kclass(name)
        if pred and not pred(name):
            continue
        for k, v in a.items():
            cls[key][k] += v
    out = {}
    for key, a in cls.items():
        n = a["n"]
        mean = lambda c: (a[c] / a["n_" + c]) if a.get("n_" + c) else None
        fetch, write = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        dur_us = a["dur_ns"] / n / 1e3
        mfma, gui, wavec, waitany = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE"), mean("SQ_WAVE_CYCLES"), mean("SQ_WAIT_ANY")
        d = {"dispatches_per_eval": round(n / evals, 2), "mean_duration_us_plain_launch": round(dur_us, 2),
             "fetch_bytes_per_dispatch": int(fetch * 1024 * 2) if fetch is not None else None,
             "write_bytes_per_dispatch_uncalibrated": int(write * 1024) if write is not None else None}
        if fetch is not None:
            d["hbm_gbs_counter"] = round((fetch * 1024 * 2 + (write or 0) * 1024) / (dur_us * 1e-6) / 1e9, 1)
        if mfma is not None and gui:
            d["mfma_busy"] = round(mfma / (gui / 8 * 1024), 5)
        if waitany is not None and wavec:
            d["wave_parked_share"] = round(waitany / wavec, 4)
        out[key] = d
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    acc = collect(B, N)
    res = {"source": "tools/unet_pmc.py %d %d: rocprofv3 --kernel-trace --pmc, three passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE) over "
                     "tools/unet_eval_loop.py (plain launches of the sampler's eval body); per-dispatch means" % (B, N),
           "unit_note": "fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950: the counter reports half of a wide coalesced read, MI355X_MICROARCH.md); WRITE_SIZE x 1024, "
                        "uncalibrated; hbm_gbs_counter = (fetch + write) / mean trace duration of a PLAIN launch (longer than the launch costs inside the "
                        "replayed graph: the counters' GB/s is a lower bound of what the graph sees); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                        "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
           "batch": B, "fused_conv_family": summarise(acc, N, pred=lambda k: any(s in k for s in FUSED)).get("selected"),
           "whole_eval": summarise(acc, N, pred=lambda k: True).get("selected"), "per_class": dict(sorted(summarise(acc, N).items()))}
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    print(txt)
