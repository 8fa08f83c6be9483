"""GPU experiments: backward timing with/without table scatter; fp32 atomic accumulation accuracy."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ngp_ref
from sparsefusion_amd import _lib
from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
from sparsefusion_amd.nerf.renderer import _FieldHandle
from sparsefusion_amd.gridencoder import backend, encoder
dev = "cuda:0"
# --- atomic accuracy: many adds to the same 8 rows
for B in (1000, 100000):
    offs = torch.tensor([0, 4920], dtype=torch.int32)
    x = torch.full((B, 3), 0.3717)
    g = torch.Generator().manual_seed(0)
    gy = torch.randn(1, B, 2, generator=g) * (1 + 10 * torch.rand(1, B, 2, generator=g))
    gt = torch.zeros(4920, 2, device=dev)
    backend.grid_encode_backward(gy.to(dev), x.to(dev), torch.zeros(4920, 2, device=dev), offs.to(dev), gt, B, 3, 2, 1, 0.0, 16, None, None, 1, False)
    nz = gt.abs().sum(1) > 0
    got = gt[nz].cpu().double()
    # exact: each of the 8 rows gets w_i * sum(gy)
    tot = gy.double().sum(1)[0]
    ratio = got / tot
    print(f"B={B}: rows hit {int(nz.sum())}; sum(w)={ratio.sum(0).tolist()} (exact 1.0); |terms|/|sum| = {(gy.abs().sum()/tot.abs().sum()).item():.1f}")
# --- backward timing
p = ngp_ref.init_params(bound=4, seed=1, table_std=0.5, sigma_bias=-3.0)
net = NeRFNetwork(get_default_torch_ngp_opt()); net.load_state_dict({k: p[k] for k in net.state_dict().keys()}); net = net.to(dev).train()
for label, (o, d) in {"one view 128x128": ngp_ref.circle_rays(128, view=7)}.items():
    o, d = o.to(dev), d.to(dev)
    N, T = o.shape[0], 64
    r = net.render(o[None], d[None], perturb=True, bg_color=0, shading='albedo', **vars(net.opt))
    h = _FieldHandle(net); params = [t.detach().contiguous() for t in net._field_params()]
    f = h.struct(params); lib = _lib.lib()
    # re-run forward through the C ABI to get the saved tensors
    f32 = dict(dtype=torch.float32, device=dev)
    nears, fars = torch.empty(N, **f32), torch.empty(N, **f32); zs, ss = torch.empty(N, 128, **f32), torch.empty(N, 128, **f32); rs = torch.empty(N, 128, 3, **f32)
    img, dep, ws = torch.empty(N, 3, **f32), torch.empty(N, **f32), torch.empty(N, **f32)
    wb = lib.sf_ngp_render_workspace_bytes(N, T); work = torch.empty(wb // 4, **f32)
    lin = torch.linspace(0, 1, T, device=dev); uc = torch.rand(N, T, device=dev); uf = torch.rand(N, T, device=dev)
    _lib.check(lib.sf_ngp_render_forward(C.byref(f), _lib.ptr(o), _lib.ptr(d), _lib.ptr(net.aabb_train), N, T, 0.1, _lib.ptr(lin), _lib.ptr(uc), _lib.ptr(uf), T, 0.0,
               _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(zs), _lib.ptr(ss), _lib.ptr(rs), _lib.ptr(img), _lib.ptr(dep), _lib.ptr(ws), _lib.ptr(work), wb, _lib.stream_ptr()))
    gi, gw = torch.randn(N, 3, device=dev), torch.randn(N, device=dev)
    for scatter in (True, False):
        grads = [torch.zeros_like(t) for t in params]
        gs = _lib.SfNgpFieldGrad()
        (gs.g_embeddings, gs.g_w0, gs.g_b0, gs.g_w1, gs.g_b1, gs.g_w2, gs.g_b2) = (t.data_ptr() for t in grads)
        if not scatter: gs.g_embeddings = 0
        def run():
            _lib.check(lib.sf_ngp_render_backward(C.byref(f), C.byref(gs), _lib.ptr(o), _lib.ptr(d), _lib.ptr(net.aabb_train), N, T, _lib.ptr(nears), _lib.ptr(fars),
                       _lib.ptr(zs), _lib.ptr(ss), _lib.ptr(rs), 0.0, _lib.ptr(gi), _lib.ptr(gw), 0, _lib.ptr(work), wb, _lib.stream_ptr()))
        run(); torch.cuda.synchronize(); t = time.time()
        for _ in range(5): run()
        torch.cuda.synchronize(); print(f"{label}: backward scatter={scatter}: {(time.time()-t)/5*1e3:.3f} ms")
