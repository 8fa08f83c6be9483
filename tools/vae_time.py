"""Time the canonical SD-VAE encode / decode on the HIP plan; per-op-type event breakdown via sf_plan_profile."""
import ctypes as C
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.vae import AutoencoderKL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
vae = AutoencoderKL().to(dev)
img, z = torch.rand(B, 3, 256, 256, device=dev), torch.randn(B, 4, 32, 32, device=dev)
NAMES = {1: "conv", 2: "gn_act", 7: "eltwise", 8: "memset"}
for kind, x, fn in (("enc", img, vae.encode), ("dec", z, vae.decode)):
    for _ in range(3):
        fn(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    plan = vae._plan(kind, B, dev)
    buf = (C.c_float * len(plan.ops))()
    _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), buf))
    _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), buf))
    per = {}
    for o, m in zip(plan.ops, buf):
        per.setdefault(NAMES.get(o.type, str(o.type)), [0, 0.0])
        per[NAMES.get(o.type, str(o.type))][0] += 1
        per[NAMES.get(o.type, str(o.type))][1] += m
    gflop = {"enc": 270.6, "dec": 620.0}[kind] * B
    print(f"B={B} {kind}: {ms:.3f} ms wall  ({gflop / ms:.1f} TFLOP/s)  ops={len(plan.ops)}  " +
          "  ".join(f"{k}: {v[0]}x {v[1]:.3f} ms" for k, v in per.items()), flush=True)
    slow = sorted(zip(buf, range(len(plan.ops))), reverse=True)[:8]
    for m, k in slow:
        o = plan.ops[k]
        print(f"    op {k} type {o.type} flags {o.flags}: {m:.3f} ms  i={list(o.i)[:15]}")
