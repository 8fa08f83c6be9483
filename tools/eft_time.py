"""Time the EFT pre-pass at the reference sizes: NC input views of 256^2, one 32x32 feature render x 20 depths per cached
view (BASELINE config 2: NC = 6).  Per-op-type event breakdown of the forward plan via sf_plan_profile."""
import collections, ctypes as C, math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.eft import EpipolarFeatureTransformer
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")


class Cams:                                     # plain pinhole cameras on a circle (stand-in for pytorch3d's)
    def __init__(self, n):
        a = torch.arange(n) * 0.4
        c, s, z, o = torch.cos(a), torch.sin(a), torch.zeros(n), torch.ones(n)
        self.R = torch.stack([torch.stack([c, z, -s], -1), torch.stack([z, o, z], -1), torch.stack([s, z, c], -1)], 1).to(dev)
        self.T = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous().to(dev)
    def __len__(self): return self.R.shape[0]
    def get_camera_center(self): return -torch.bmm(self.T[:, None], self.R.transpose(1, 2))[:, 0]
    def transform_points_ndc(self, p):
        cam = torch.bmm(p.expand(len(self), -1, -1), self.R) + self.T[:, None]
        return torch.cat([cam[..., :2] / cam[..., 2:3] * 2.2, 1 / cam[..., 2:3]], -1)


RB = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])
eft = EpipolarFeatureTransformer(use_r=True, encoder='resnet18', return_features=True, remove_unused_layers=False).to(dev)
cams, rgb = Cams(NC), torch.rand(NC, 3, 256, 256, device=dev)
N, D = 32 * 32, 20
ax = torch.linspace(-0.5, 0.5, 32)
yy, xx = torch.meshgrid(ax, ax, indexing="ij")
d = torch.stack([xx.reshape(-1), yy.reshape(-1), torch.ones(N)], -1).to(dev)
o = torch.tensor([0.3, 0.1, -4.0], device=dev).expand(N, 3).contiguous()
lengths = torch.linspace(1.5, 6.5, D, device=dev).expand(N, D).contiguous()
rb = RB(o, d, lengths, None)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


t_enc = timed(lambda: eft.encode(cams, rgb))
t_fwd = timed(lambda: eft(rb))
print(f"NC={NC}: encode {t_enc:.3f} ms (resnet18 trunk on {NC} x 256^2), forward {t_fwd:.3f} ms "
      f"({N} rays x {D} depths = {NC * N * D} tokens), per cached view {t_enc + t_fwd:.3f} ms")
plan = [v for k, v in eft._plans.items() if k[0] == "fwd"][0]
buf = (C.c_float * len(plan.ops))()
for _ in range(2):
    _lib.check(_lib.lib().sf_plan_profile(plan.op_array, len(plan.ops), _lib.stream_ptr(), buf))
names = {1: "conv/linear", 3: "layernorm", 13: "eft ops"}
per = {}
for op, m in zip(plan.ops, buf):
    k = names.get(op.type, str(op.type)) + (f"[{op.flags}]" if op.type == 13 else "")
    per.setdefault(k, [0, 0.0]); per[k][0] += 1; per[k][1] += m
print("  forward plan:", "  ".join(f"{k}: {v[0]}x {v[1]:.3f} ms" for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])))
flops = 2 * NC * N * D * (608 * 256 + 448 * 256 + 4 * 2 * (768 * 256 + 3 * 256 * 256)) / 1e9
print(f"  linear layers of T1+T2: {flops:.1f} GFLOP")
