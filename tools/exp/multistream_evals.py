"""K independent B = 1 UNet evals on K streams vs the same K views as one B = K eval (r05).
A B = 1 eval is a chain of ~126 dependent launches that keeps no unit of the chip busy more than a quarter of the time; independent
trajectories (the K novel views of BASELINE configs[3]) could interleave their chains instead of widening every launch.
usage: multistream_evals.py [K] [rounds]"""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsefusion_amd.unet import Unet

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
mk = lambda: Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                  layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
nets = [mk() for _ in range(K)]
ls = torch.linspace(-3, 3, 8, device=dev)
ctxs = [n.begin_sampling(torch.randn(1, 256, 32, 32, device=dev), ls) for n in nets]
xs = [torch.randn(1, 4, 32, 32, device=dev) for _ in range(K)]
for n, c, x in zip(nets, ctxs, xs):
    n.eval_prepared(c, x, 0)                      # captures the body graph
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(K)]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / R * 1e3


def sequential():
    for n, c, x in zip(nets, ctxs, xs):
        n.eval_prepared(c, x, 1, row_ready=True)


def concurrent():
    for s, n, c, x in zip(streams, nets, ctxs, xs):
        with torch.cuda.stream(s):
            n.eval_prepared(c, x, 1, row_ready=True)


t_seq = timed(sequential)
t_con = timed(concurrent)
cb = nets[0].begin_sampling(torch.randn(K, 256, 32, 32, device=dev), ls)
xb = torch.randn(K, 4, 32, 32, device=dev)
nets[0].eval_prepared(cb, xb, 0)
t_bat = timed(lambda: nets[0].eval_prepared(cb, xb, 1, row_ready=True))
print(f"K = {K}: {K} B=1 evals one after the other {t_seq:.3f} ms | on {K} streams {t_con:.3f} ms | one B={K} eval {t_bat:.3f} ms")
