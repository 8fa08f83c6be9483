"""`_raymarching` backend on libsparsefusion_hip.so: same positional signatures as the
reference pybind module (raymarching/src/bindings.cpp:7-18, raymarching.h:7-18) for the
entry points the distillation path uses.  All outputs are caller-allocated."""
import torch

from .. import _lib


def _f32(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous float32 tensor")


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (aabb, "aabb"), (nears, "nears"), (fars, "fars")):
        _f32(t, n)
    rc = _lib.lib().sf_near_far_from_aabb(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), int(N),
                                          float(min_near), _lib.ptr(nears), _lib.ptr(fars), _lib.stream_ptr())
    _lib.check(rc, "near_far_from_aabb")


def morton3D(coords, N, indices):
    _lib.require_cuda(coords, indices)
    assert coords.dtype == torch.int32 and indices.dtype == torch.int32
    _lib.check(_lib.lib().sf_morton3D(_lib.ptr(coords.contiguous()), int(N), _lib.ptr(indices), _lib.stream_ptr()),
               "morton3D")


def morton3D_invert(indices, N, coords):
    _lib.require_cuda(coords, indices)
    assert coords.dtype == torch.int32 and indices.dtype == torch.int32
    _lib.check(_lib.lib().sf_morton3D_invert(_lib.ptr(indices.contiguous()), int(N), _lib.ptr(coords),
                                             _lib.stream_ptr()), "morton3D_invert")


def packbits(grid, N, density_thresh, bitfield):
    _f32(grid, "grid")
    _lib.require_cuda(bitfield)
    assert bitfield.dtype == torch.uint8
    _lib.check(_lib.lib().sf_packbits(_lib.ptr(grid), int(N), float(density_thresh), _lib.ptr(bitfield),
                                      _lib.stream_ptr()), "packbits")
