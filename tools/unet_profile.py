"""Per-op HIP-event profile of one UNet eval (canonical config): per-type totals and the slowest ops."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.unet import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NAMES = {14: "FCONV", 15: "SLOTS", 16: "GCA", 10: "SPLITK_RED", 1: "CONV", 2: "GN_ACT", 3: "LN", 4: "GEMV", 5: "ATTN", 6: "GCA_POOL", 7: "ELTWISE", 8: "MEMSET", 9: "TIME_EMB"}
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
for target in [int(a) for a in sys.argv[2:]] or [512]:
    unet.conv_waves_target = target
    unet._plans = {}
    plan = unet._plan(B, dev)
    ms = (C.c_float * len(plan.ops))()
    acc = np.zeros(len(plan.ops))
    for it in range(6):
        _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), ms))
        if it:
            acc += np.array(list(ms))
    acc /= 5
    print(f"== B={B} conv_waves_target={target}: {len(plan.ops)} ops, {acc.sum():.3f} ms per eval (event-serialised)")
    tot = {}
    for o, m in zip(plan.ops, acc):
        tot.setdefault(o.type, [0, 0.0]); tot[o.type][0] += 1; tot[o.type][1] += m
    for t, (n, m) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"   {NAMES[t]:9s} n={n:3d} total {m:7.3f} ms  avg {m / n * 1e3:7.1f} us")
    order = np.argsort(-acc)[:14]
    for k in order:
        o = plan.ops[k]
        print(f"   op{k:3d} {NAMES[o.type]:8s} {acc[k] * 1e3:7.1f} us  i={list(o.i)[:15]} flags={o.flags}")
    # wall time without events
    torch.cuda.synchronize()
    import time
    t = time.time()
    for _ in range(20):
        _lib.check(_lib.lib().sf_plan_run(plan.op_array, len(plan.ops), _lib.stream_ptr()))
    torch.cuda.synchronize()
    print(f"   plain sf_plan_run wall: {(time.time() - t) / 20 * 1e3:.3f} ms per eval")
    x, ls, cond = torch.randn(B, 4, 32, 32, device=dev), torch.zeros(B, device=dev), torch.randn(B, 256, 32, 32, device=dev)
    unet.forward(x, ls, cond_images=cond); torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        unet.forward(x, ls, cond_images=cond)
    torch.cuda.synchronize()
    print(f"   Unet.forward (hipGraph replay + I/O copies) wall: {(time.time() - t) / 20 * 1e3:.3f} ms per eval")
