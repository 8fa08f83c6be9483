"""Join a rocprofv3 kernel trace of tools/unet_eval_loop.py with the UNet op plan: per-conv duration, bytes, GB/s."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd.unet import Unet, _Plan
import sparsefusion_amd.unet as U
import torch
trace = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(trace)))
idx = [i for i, r in enumerate(rows) if 'k_pack_in' in r['Kernel_Name']]
seg = rows[idx[-1]:]
# plan (sizing pass needs no GPU): monkeypatch packed weights to dummy pointers
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
class FakeW(dict):
    def __getitem__(self, k):
        class P:
            def data_ptr(self): return 4096
        return P()
unet._packed = lambda dev: FakeW()
plan = _Plan(unet, 1, None).build()
convs = [o for o in plan.ops if o.type == 1]
ktr = [r for r in seg if 'k_conv_igemm' in r['Kernel_Name']]
assert len(convs) == len(ktr), (len(convs), len(ktr))
tot = 0
print("  M     N    K(in*taps) tile grp   us    wMB   GB/s")
for o, r in zip(convs, ktr):
    i = list(o.i)
    B, H, W, Cin, Ho, Wo, Cout, ldc, co, kh, kw, st, pad, g, tile = i[:15]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    wb = Cout * Cin * kh * kw * 2
    tot += d
    print(f"{B*Ho*Wo:5d} {Cout:5d} {Cin:5d}x{kh*kw:<3d} {tile//16}x{tile%16} {g:3d} {d:6.1f} {wb/1e6:6.2f} {wb/d/1e3:7.1f}")
print("conv total us", tot, "all kernels us", sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in seg) / 1e3, "n", len(seg))
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r['Kernel_Name'].split('(')[0][:28]
    agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:30s} n={c:3d} {t:8.1f} us  avg {t/c:5.1f}")
