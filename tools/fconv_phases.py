"""Phase breakdown of k_conv_fused on the UNet's own layer shapes: per-workgroup timestamps (100 MHz clock) taken inside the
kernel (FConvArgs.dbg) -> microseconds spent in weight-prefetch issue, statistics, staging, main loop, epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ctypes as C
import fused_cases as fc
from sparsefusion_amd import build

fc.TIMING_LIB = C.CDLL(os.environ.get("SF_TIMING_LIB") or build.build_timing(verbose=False))
fc.TIMING_LIB.sf_fused_op_run.restype = C.c_int

names = sys.argv[1:] or sorted(fc.CONV_CASES_FULL)
for name in names:
    attn = name.startswith("attn_")                             # attn_self / attn_cross: the attention prologue + output projection (1024 channels)
    kw = dict(B=1, cross=name == "attn_cross", Cout=1024, seed=64) if attn else dict(fc.CONV_CASES_FULL[name])
    if kw.get("accum"):
        continue
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda:0")
    try:
        wall = fc.run_attn_case("gpu", **kw, dbg=dbg, reps=20) if attn else fc.run_conv_case("gpu", **kw, dbg=dbg, reps=20)
    except Exception as e:                                       # paired launches carry no stamps
        print(f"{name:28s} skipped ({e})")
        continue
    d = dbg.view(-1, 8).cpu()
    d = d[d[:, 5] != 0].double()
    if d.shape[0] == 0:                                        # a library without the stamps (counter runs)
        print(f"{name:28s} wall/launch {wall * 1e6:6.1f} us (no phase stamps in this build)")
        continue
    t0 = d[:, 0].min()
    ph = (d[:, 1:6] - d[:, 0:5]) / 100.0                      # us per phase per workgroup
    start = (d[:, 0] - t0) / 100.0
    end = (d[:, 5] - t0) / 100.0
    print(f"{name:28s} WGs {d.shape[0]:4d}  wall/launch {wall * 1e6:6.1f} us  kernel span {float(end.max()):6.2f} us  "
          f"WG start spread {float(start.max()):5.2f}")
    pipe = bool(kw.get("pipe"))       # k_conv_fused_pipe stamps: 1 = slot statistics + table done, 2 = first chunk staged, 3 = 4 = pipeline drained
    for k, lab in enumerate(["entry->statistics", "first chunk staged", "pipelined chunks", "(-)", "epilogue"] if pipe else
                            ["prefetch-issue", "statistics", "staging", "main loop", "epilogue"]):
        print(f"      {lab:15s} mean {float(ph[:, k].mean()):6.2f}  max {float(ph[:, k].max()):6.2f}")
    if float(d[:, 6].max()) > 0:                               # finer stamps inside "prefetch-issue": entry -> s6 -> s7 -> stamp 1
        a6, a7 = (d[:, 6] - d[:, 0]) / 100.0, (d[:, 7] - d[:, 6]) / 100.0
        a1 = (d[:, 1] - d[:, 7]) / 100.0
        print(f"      [entry->first loads issued {float(a6.mean()):5.2f} | ->second block {float(a7.mean()):5.2f} | ->stamp1 {float(a1.mean()):5.2f}]")
