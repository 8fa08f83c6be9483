"""SF_OP_INITX alone: hot (back-to-back launches) vs cold (a 512 MB fill between launches evicts L2 / MALL) time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsefusion_amd import _lib
from sparsefusion_amd.unet import init_x_weight_table

DEV = "cuda:0"
B, R, Cx, cws = 1, 32, 4, (128, 64, 64)
g = torch.Generator().manual_seed(0)
ws = [torch.randn(cw, Cx, k, k, generator=g) for cw, k in zip(cws, (3, 7, 15))]
tab, woffs = init_x_weight_table(ws)
x = torch.randn(B, Cx, R, R, device=DEV)
base = torch.randn(B * R * R, 256, device=DEV)
out = torch.empty_like(base)
tab = tab.to(DEV)
o = _lib.SfOp()
o.type, o.flags = 17, 0
for k, v in enumerate((x, base, tab, out)):
    o.p[k] = v.data_ptr()
for k, v in enumerate((B, R, R, Cx, 256) + cws + (0, 128, 192) + tuple(woffs)):
    o.i[k] = int(v)
arr = (_lib.SfOp * 1)(o)
lib = _lib.lib()


def launch():
    _lib.check(lib.sf_plan_run(arr, 1, _lib.stream_ptr()), "plan")


for _ in range(5):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    launch()
e1.record(); torch.cuda.synchronize()
print(f"hot : {e0.elapsed_time(e1) / 200 * 1e3:.2f} us / launch")
big = torch.empty(128 << 20, device=DEV)
ts = []
for _ in range(20):
    big.fill_(1.0)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"cold: {sorted(ts)[len(ts) // 2]:.2f} us / launch (event pair around one launch)")
