"""Is the ~10 us fixed cost of a weight-streaming conv a cold-miss (TLB / MALL) effect?  Same conv, warm vs cold weights."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sparsefusion_amd import _lib
from test_gpu_unet_ops import _op, DEV
lib = _lib.lib()
def packed(co, ci, k):
    n = lib.sf_conv_packed_elems(co, ci, k, k)
    return torch.randint(-3000, 3000, (n,), dtype=torch.int16, device=DEV)
def bench(ops, reps=20):
    arr = (_lib.SfOp * len(ops))(*ops)
    ms = (C.c_float * len(ops))()
    acc = np.zeros(len(ops))
    for it in range(reps + 2):
        _lib.check(lib.sf_plan_profile(arr, len(ops), _lib.stream_ptr(), ms))
        if it >= 2: acc += np.array(list(ms))
    return acc / reps * 1e3
B, H, Cin, Cout, k = 1, 4, 1024, 1024, 3
x = torch.randn(B, H, H, Cin, device=DEV).to(torch.bfloat16)
out = torch.zeros(B, H, H, Cout, device=DEV)
ws = torch.empty(32 * 16 * Cout, device=DEV)
nW = 24
Ws = [packed(Cout, Cin, k) for _ in range(nW)]     # 24 x 18.9 MB = 453 MB > MALL
junk = torch.empty(300 * 1024 * 1024 // 4, device=DEV)
for g, tile in ((4, 17), (8, 17), (2, 17)):
    mk = lambda w: _op(1, 0, p=(x, w, None, out, None, ws), i=(B, H, H, Cin, H, H, Cout, Cout, 0, k, k, 1, 1, g, tile))
    warm = bench([mk(Ws[0])] * 8)
    cold = bench([mk(w) for w in Ws])
    print(f"groups={g}: same weights back-to-back {warm.mean():.1f} us (min {warm.min():.1f}); cycling 24 different weight sets {cold.mean():.1f} us  [event-timed, ~5us event overhead each]")
# tiny kernel floor for reference
t = torch.zeros(2, device=DEV); w8 = torch.randn(8, device=DEV); emb = torch.empty(2, 17, device=DEV)
print("time_emb floor", bench([_op(9, 0, p=(t, w8, None, emb), i=(2, 8))] * 8).mean())
