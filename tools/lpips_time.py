"""Per-op event profile of the LPIPS forward / backward plans at 256^2 (LDS-tiled conv on / off)."""
import ctypes as C
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.lpips import LPIPS
dev = torch.device("cuda:0")
for lds in (0, 96):
    net = LPIPS().to(dev)
    net.lds_conv_min_blocks = lds
    a = torch.rand(1, 3, 256, 256, device=dev, requires_grad=True)
    b = torch.rand(1, 3, 256, 256, device=dev)
    net(a, b).sum().backward()
    torch.cuda.synchronize()
    fwd, bwd = net._plans_for(1, 256, dev)
    for name, plan in (("fwd", fwd), ("bwd", bwd)):
        buf = (C.c_float * len(plan.ops))()
        for _ in range(3):
            _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), buf))
        tot = sum(buf)
        print(f"lds_min={lds} {name}: {tot:.3f} ms, {len(plan.ops)} ops")
        for k, (o, m) in enumerate(zip(plan.ops, buf)):
            if m > 0.08:
                print(f"    op {k} type {o.type} flags {o.flags}: {m:.3f} ms  i={list(o.i)[:15]}")
