"""GPU diagnostic: isolated backward vs oracle, per parameter / per level, and the intermediate dsigma/drgb."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import ngp_ref
from ngp_common import params_from_cfg, grad_leaf
from sparsefusion_amd import _lib
from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
from sparsefusion_amd.nerf.renderer import _FieldHandle
dev = "cuda:0"
g = torch.load("tests/golden/ngp_render.pt")["teacher"]
p = params_from_cfg(g["cfg"]); pl = grad_leaf(p)
ref = ngp_ref.render_run(pl, g["rays_o"], g["rays_d"], u_coarse=g["u_coarse"], u_fine=g["u_fine"], bg_color=0.0, training=True, return_aux=True)
ref["sigma_sorted"].retain_grad(); ref["rgb_sorted"].retain_grad()
((ref["image"] * g["g_image"]).sum() + (ref["weights_sum"] * g["g_ws"]).sum()).backward()
net = NeRFNetwork(get_default_torch_ngp_opt()); net.load_state_dict({k: p[k] for k in net.state_dict().keys()}); net = net.to(dev)
h = _FieldHandle(net); params = [t.detach().contiguous() for t in net._field_params()]
grads = [torch.zeros_like(t) for t in params]; f = h.struct(params); gs = _lib.SfNgpFieldGrad()
(gs.g_embeddings, gs.g_w0, gs.g_b0, gs.g_w1, gs.g_b1, gs.g_w2, gs.g_b2) = (t.data_ptr() for t in grads)
N, T = 256, 64
d = lambda t: t.detach().contiguous().to(dev)
o, dd, aabb = d(g["rays_o"]), d(g["rays_d"]), d(p["aabb_train"])
nears, fars, zs, ss, rs = d(ref["nears"]), d(ref["fars"]), d(ref["z_sorted"]), d(ref["sigma_sorted"]), d(ref["rgb_sorted"])
gi, gw = d(g["g_image"]), d(g["g_ws"])
lib = _lib.lib(); wb = lib.sf_ngp_render_workspace_bytes(N, T); work = torch.zeros(wb // 4, device=dev)
_lib.check(lib.sf_ngp_render_backward(C.byref(f), C.byref(gs), _lib.ptr(o), _lib.ptr(dd), _lib.ptr(aabb), N, T, _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(zs), _lib.ptr(ss), _lib.ptr(rs), 0.0, _lib.ptr(gi), _lib.ptr(gw), 0, _lib.ptr(work), wb, _lib.stream_ptr()))
torch.cuda.synchronize()
dsig = work[:N * 128].view(N, 128).cpu(); drgb = work[N * 128:N * 128 * 4].view(N, 128, 3).cpu()
# sigma gradient of the oracle: sigma_sorted is used by weights only; rgb_sorted by image only
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("dsig rel", rel(dsig, ref["sigma_sorted"].grad), "drgb rel", rel(drgb, ref["rgb_sorted"].grad))
names = ["encoder.embeddings"] + [f"sigma_net.net.{i}.{w}" for i in range(3) for w in ("weight", "bias")]
for n, got in zip(names, grads):
    print(n, "rel", rel(got.cpu(), pl[n].grad))
offs = p["encoder.offsets"]; ge = grads[0].cpu(); want = pl["encoder.embeddings"].grad
for l in range(16):
    print("level", l, rel(ge[offs[l]:offs[l+1]], want[offs[l]:offs[l+1]]))
