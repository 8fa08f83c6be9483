"""ctypes binding of oracle/ngp_ref.c.  Test infrastructure only (see oracle/__init__.py).

Functions take/return CPU torch tensors (fp32 / int32) and mirror the positional signatures
of the reference's pybind modules so the reference's own Python (loaded by ref_loader.py) can
run on top of them."""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ngp.so")
_SO_NOFMA = os.path.join(_HERE, "_build", "liboracle_ngp_nofma.so")
_lib = None
_libs = {}


def build(force=False):
    src = os.path.join(_HERE, "ngp_ref.c")
    if force or not os.path.exists(_SO) or not os.path.exists(_SO_NOFMA) or min(os.path.getmtime(_SO), os.path.getmtime(_SO_NOFMA)) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _libs.setdefault(_SO, C.CDLL(_SO))
    return _lib


@contextlib.contextmanager
def unfused():
    """Inside this context every function of this module runs on the -DORACLE_NO_FMA build (all fmaf() unfused): the form that
    is compared bit for bit with the reference's sources compiled -ffp-contract=off (ref_native.unfused())."""
    global _lib
    build()
    prev = _lib
    _lib = _libs.setdefault(_SO_NOFMA, C.CDLL(_SO_NOFMA))
    try:
        yield
    finally:
        _lib = prev


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, dtype=torch.float32):
    assert t.device.type == "cpu" and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C_, L, S, H, dy_dx, gridtype, align_corners,
                        level_index=None):
    _chk(inputs); _chk(embeddings); _chk(outputs); _chk(offsets, torch.int32)
    lib().oracle_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), C.c_uint32(B),
                                     C.c_uint32(D), C.c_uint32(C_), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                     _p(dy_dx), C.c_uint32(gridtype), C.c_int(int(bool(align_corners))),
                                     _p(level_index))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C_, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners):
    _chk(grad); _chk(inputs); _chk(grad_embeddings); _chk(offsets, torch.int32)
    lib().oracle_grid_encode_backward(_p(grad), _p(inputs), _p(offsets), _p(grad_embeddings), C.c_uint32(B),
                                      C.c_uint32(D), C.c_uint32(C_), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                      _p(dy_dx), _p(grad_inputs), C.c_uint32(gridtype),
                                      C.c_int(int(bool(align_corners))))


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    _chk(rays_o); _chk(rays_d); _chk(aabb); _chk(nears); _chk(fars)
    lib().oracle_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), C.c_uint32(N), C.c_float(min_near),
                                    _p(nears), _p(fars))


def morton3D(coords, N, indices):
    _chk(coords, torch.int32); _chk(indices, torch.int32)
    lib().oracle_morton3D(_p(coords), C.c_uint32(N), _p(indices))


def morton3D_invert(indices, N, coords):
    _chk(coords, torch.int32); _chk(indices, torch.int32)
    lib().oracle_morton3D_invert(_p(indices), C.c_uint32(N), _p(coords))


def packbits(grid, N, thresh, bitfield):
    _chk(grid); _chk(bitfield, torch.uint8)
    lib().oracle_packbits(_p(grid), C.c_uint32(N), C.c_float(thresh), _p(bitfield))


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    _chk(rays_o); _chk(rays_d); _chk(coords)
    lib().oracle_sph_from_ray(_p(rays_o), _p(rays_d), C.c_float(radius), C.c_uint32(N), _p(coords))


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C_, H, M, nears, fars, xyzs, dirs, deltas, rays,
                     counter, noises):
    for t in (rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises):
        _chk(t)
    _chk(grid, torch.uint8); _chk(rays, torch.int32); _chk(counter, torch.int32)
    lib().oracle_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), C.c_float(bound), C.c_float(dt_gamma),
                                  C.c_uint32(max_steps), C.c_uint32(N), C.c_uint32(C_), C.c_uint32(H), C.c_uint32(M),
                                  _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), _p(noises))


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    for t in (sigmas, rgbs, deltas, weights_sum, depth, image):
        _chk(t)
    _chk(rays, torch.int32)
    lib().oracle_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), C.c_uint32(M), C.c_uint32(N),
                                              C.c_float(T_thresh), _p(weights_sum), _p(depth), _p(image))


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                                  grad_sigmas, grad_rgbs):
    for t in (grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, grad_sigmas, grad_rgbs):
        _chk(t)
    _chk(rays, torch.int32)
    lib().oracle_composite_rays_train_backward(_p(grad_weights_sum), _p(grad_image), _p(sigmas), _p(rgbs), _p(deltas),
                                               _p(rays), _p(weights_sum), _p(image), C.c_uint32(M), C.c_uint32(N),
                                               C.c_float(T_thresh), _p(grad_sigmas), _p(grad_rgbs))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C_, H, grid, nears, fars,
               xyzs, dirs, deltas, noises):
    for t in (rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises):
        _chk(t)
    _chk(grid, torch.uint8); _chk(rays_alive, torch.int32)
    lib().oracle_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                            C.c_float(bound), C.c_float(dt_gamma), C.c_uint32(max_steps), C.c_uint32(C_), C.c_uint32(H),
                            _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises))


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    for t in (rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        _chk(t)
    _chk(rays_alive, torch.int32)
    lib().oracle_composite_rays(C.c_uint32(n_alive), C.c_uint32(n_step), C.c_float(T_thresh), _p(rays_alive), _p(rays_t),
                                _p(sigmas), _p(rgbs), _p(deltas), _p(weights_sum), _p(depth), _p(image))


class GridEncodeCPU(torch.autograd.Function):
    """Autograd glue around the C oracle with the data flow of external/gridencoder/grid.py:19-88
    ([L,B,C] kernel layout, permuted to [B, L*C]; zero-initialised table gradient)."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, align_corners=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        Cc = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        out = torch.empty(L, B, Cc)
        dy_dx = torch.empty(B, L * D * Cc) if calc_grad_inputs else None
        grid_encode_forward(inputs, embeddings.contiguous(), offsets, out, B, D, Cc, L, S, base_resolution, dy_dx,
                            gridtype, align_corners)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, Cc, L, S, base_resolution, gridtype, align_corners)
        return out.permute(1, 0, 2).reshape(B, L * Cc)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cc, L, S, H, gridtype, ac = ctx.dims
        g = grad.view(B, L, Cc).permute(1, 0, 2).contiguous()
        ge = torch.zeros_like(embeddings)
        gi = torch.zeros_like(inputs) if dy_dx is not None else None
        grid_encode_backward(g, inputs, embeddings.contiguous(), offsets, ge, B, D, Cc, L, S, H, dy_dx, gi, gridtype, ac)
        return gi, ge, None, None, None, None, None, None
