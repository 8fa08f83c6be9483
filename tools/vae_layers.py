"""Per-layer table of the SD-VAE encode / decode plans: every conv op with its shape, tile, event-timed duration and TFLOP/s."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.vae import AutoencoderKL
dev = torch.device("cuda:0")
vae = AutoencoderKL().to(dev)
img, z = torch.rand(1, 3, 256, 256, device=dev), torch.randn(1, 4, 32, 32, device=dev)
for kind, x, fn in (("enc", img, vae.encode), ("dec", z, vae.decode)):
    for _ in range(3):
        fn(x)
    plan = vae._plan(kind, 1, dev)
    buf = (C.c_float * len(plan.ops))()
    acc = [0.0] * len(plan.ops)
    for it in range(4):
        _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), buf))
        if it:
            acc = [a + b for a, b in zip(acc, buf)]
    agg = {}
    for o, m in zip(plan.ops, acc):
        if o.type != 1:
            continue
        B, H, W, Cin, Ho, Wo, Cout, k, tile = o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.i[9], o.i[14]
        key = (H, W, Cin, Ho, Wo, Cout, k, tile, o.flags & 1)
        agg.setdefault(key, [0, 0.0])
        agg[key][0] += 1
        agg[key][1] += m / 3
    tot = sum(m for m in acc) / 3
    print(f"== {kind}: plan {tot:.3f} ms event-timed per op; conv layers (count, ms total, TFLOP/s)")
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        H, W, Cin, Ho, Wo, Cout, k, tile, f32 = key
        fl = 2.0 * Ho * Wo * Cout * Cin * k * k * n
        print(f"  {H:3d}x{W:<3d} {Cin:4d}->{Cout:4d} k{k} out {Ho}x{Wo} tile {tile:3d} a_f32={f32}  x{n:2d}  {ms:6.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
