"""Event-timed single conv layers on the LDS-tiled kernels: python tools/conv_time.py [sel ...]  (sel 1 = k_conv_lds, 3 / 4 = k_conv_glds
ring depth, 6 / 7 = k_conv3_halo)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
dev = torch.device("cuda:0")
sels = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 4]
SHAPES = [(128, 256, 256, 8), (64, 512, 512, 4), (256, 128, 128, 8), (128, 512, 256, 8), (32, 512, 512, 4)]   # H, Cin, Cout, bnf
if any(a.startswith('--shape=') for a in sys.argv):
    SHAPES = [SHAPES[int(a.split('=')[1])] for a in sys.argv if a.startswith('--shape=')]
lib = _lib.lib()
for H, Cin, Cout, bnf in SHAPES:
    x = torch.randn(1, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    buf = torch.empty(lib.sf_conv_packed_elems(Cout, Cin, 3, 3), dtype=torch.int16)
    _lib.check(lib.sf_conv_pack_weights(w.contiguous().data_ptr(), Cout, Cin, Cin, 3, 3, buf.data_ptr()))
    wp, bias, out = buf.to(dev), torch.zeros(Cout, device=dev), torch.empty(1, H, H, Cout, device=dev)
    row = f"{H:3d}x{H:<3d} {Cin:3d}->{Cout:3d} bnf {bnf}:"
    for sel in sels:
        o = _lib.SfOp()
        o.type, o.flags = 1, 0
        for k, v in enumerate((x, wp, bias, out)):
            o.p[k] = v.data_ptr()
        for k, v in enumerate((1, H, H, Cin, H, H, Cout, Cout, 0, 3, 3, 1, 1, 1, 256 + 16 * sel + bnf)):
            o.i[k] = v
        arr = (_lib.SfOp * 1)(o)
        run = lambda: _lib.check(lib.sf_plan_run(arr, 1, _lib.stream_ptr()))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        row += f"  sel {sel}: {us:7.1f} us {2.0 * H * H * Cin * Cout * 9 / us / 1e6:6.1f} TF"
    print(row, flush=True)
