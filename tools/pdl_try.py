"""Next-round experiment: the software-dependent-launch variant of the library on the real UNet eval.
Builds libsparsefusion_hip_pdl.so (-DSF_PDL=1), then in child processes (the library is chosen at import) runs one eval
through the default library and through the variant, compares the outputs, prints both eval times and the number of hand-off
waits that gave up (must be 0).   python tools/pdl_try.py          (on the GPU box; ~1 min)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, time, torch, ctypes
sys.path.insert(0, %(root)r)
from sparsefusion_amd import _lib
from sparsefusion_amd.unet import Unet
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
           layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
with torch.no_grad():
    net.final_conv.weight.normal_(0, 0.02)
g = torch.Generator().manual_seed(1)
x = torch.randn(1, 4, 32, 32, generator=g).to(dev); cond = torch.randn(1, 256, 32, 32, generator=g).to(dev)
ctx = net.begin_sampling(cond, torch.linspace(-3, 3, 8, device=dev))
y = net.eval_prepared(ctx, x, 3).clone()
torch.cuda.synchronize()
t = time.time()
for k in range(100):
    net.eval_prepared(ctx, x, k %% 8)
torch.cuda.synchronize()
ms = (time.time() - t) * 10
to = -1
try:
    to = _lib.lib().sf_pdl_timeouts()
except AttributeError:
    pass
torch.save({'y': y.cpu(), 'ms': ms, 'timeouts': to}, %(out)r)
"""

if __name__ == "__main__":
    from sparsefusion_amd import build
    lib = build.build_variant("pdl", ["SF_PDL=1"], verbose=False, sources=("unet_fused.hip", "unet_ops.hip"))
    import torch
    res = {}
    for tag, env in (("default", {}), ("pdl", {"SF_HIP_LIB": lib}), ("pdl lib, SF_PDL=0", {"SF_HIP_LIB": lib, "SF_PDL": "0"})):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "o.pt")
            try:
                subprocess.check_call([sys.executable, "-c", CHILD % {"root": ROOT, "out": out}], env=dict(os.environ, **env), timeout=120)
                res[tag] = torch.load(out)
            except Exception as e:                                   # a hang is cut off by the timeout
                print(f"{tag}: FAILED ({e})")
    for tag, r in res.items():
        d = float((r["y"] - res["default"]["y"]).norm() / res["default"]["y"].norm()) if "default" in res else float("nan")
        print(f"{tag:20s} eval {r['ms']:.3f} ms   rel diff vs default {d:.2e}   hand-off timeouts {r['timeouts']}")
