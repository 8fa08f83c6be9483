"""ctypes harness for tests/hostemu/ngp_host.cpp (host build of the device functions)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libngp_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "ngp_host.cpp")
        hdr = os.path.join(_HERE, "..", "..", "sparsefusion_amd", "csrc", "ngp_device.h")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-mfma",
                                   "-mavx2", "-fopenmp", "-Wno-unknown-pragmas", src, "-o", _SO])
        _lib = C.CDLL(_SO)
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _field_args(params, bound, S):
    offs = params["encoder.offsets"].contiguous()
    return [_p(params["encoder.embeddings"]), _p(offs), C.c_uint32(offs.numel() - 1), C.c_float(S), C.c_uint32(16),
            C.c_uint32(1)] + [_p(params[f"sigma_net.net.{i}.{w}"].contiguous()) for i in range(3) for w in ("weight", "bias")] \
        + [C.c_float(bound)]


def render_forward(params, rays_o, rays_d, aabb, T, min_near, bound, S, u_coarse, u_fine, bg):
    N = rays_o.shape[0]
    lin = torch.linspace(0.0, 1.0, T)
    if u_fine is None:
        u_fine_t, stride = torch.linspace(0. + 0.5 / T, 1. - 0.5 / T, steps=T).contiguous(), 0
    else:
        u_fine_t, stride = u_fine.contiguous(), T
    out = dict(nears=torch.empty(N), fars=torch.empty(N), z_sorted=torch.empty(N, 2 * T), sigma_s=torch.empty(N, 2 * T),
               rgb_s=torch.empty(N, 2 * T, 3), image=torch.empty(N, 3), depth=torch.empty(N), weights_sum=torch.empty(N),
               z_fine=torch.empty(N, T))
    lib().emu_render_forward(*_field_args(params, bound, S), _p(rays_o.contiguous()), _p(rays_d.contiguous()),
                             _p(aabb.contiguous()), C.c_uint32(N), C.c_uint32(T), C.c_float(min_near), _p(lin),
                             _p(u_coarse.contiguous() if u_coarse is not None else None), _p(u_fine_t),
                             C.c_uint32(stride), C.c_float(bg), _p(out["nears"]), _p(out["fars"]), _p(out["z_sorted"]),
                             _p(out["sigma_s"]), _p(out["rgb_s"]), _p(out["image"]), _p(out["depth"]),
                             _p(out["weights_sum"]), _p(out["z_fine"]))
    return out


def render_backward(params, rays_o, rays_d, aabb, T, bound, S, fwd, bg, g_image, g_ws):
    N = rays_o.shape[0]
    grads = {k: torch.zeros_like(params[k]) for k in ["encoder.embeddings"] +
             [f"sigma_net.net.{i}.{w}" for i in range(3) for w in ("weight", "bias")]}
    lib().emu_render_backward(*_field_args(params, bound, S), _p(rays_o.contiguous()), _p(rays_d.contiguous()),
                              _p(aabb.contiguous()), C.c_uint32(N), C.c_uint32(T), _p(fwd["nears"]), _p(fwd["fars"]),
                              _p(fwd["z_sorted"]), _p(fwd["sigma_s"]), _p(fwd["rgb_s"]), C.c_float(bg),
                              _p(g_image.contiguous()), _p(g_ws.contiguous() if g_ws is not None else None),
                              _p(grads["encoder.embeddings"]),
                              *[_p(grads[f"sigma_net.net.{i}.{w}"]) for i in range(3) for w in ("weight", "bias")])
    return grads
