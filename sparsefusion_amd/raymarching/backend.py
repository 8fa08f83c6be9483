"""`_raymarching` backend on libsparsefusion_hip.so: same positional signatures as the
reference pybind module (raymarching/src/bindings.cpp:7-18, raymarching.h:7-18) for the
entry points the distillation path uses and for the occupancy-grid (cuda_ray=True) ones.
All outputs are caller-allocated and mutated in place exactly where the reference mutates them."""
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


def _i32(t, name):
    if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous int32 CUDA tensor")


def _u8(t, name):
    if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous uint8 CUDA tensor")


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (coords, "coords")):
        _f32(t, n)
    _lib.check(_lib.lib().sf_sph_from_ray(_lib.ptr(rays_o), _lib.ptr(rays_d), float(radius), int(N), _lib.ptr(coords),
                                          _lib.stream_ptr()), "sph_from_ray")


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                     counter, noises):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (nears, "nears"), (fars, "fars"), (xyzs, "xyzs"), (dirs, "dirs"),
                 (deltas, "deltas"), (noises, "noises")):
        _f32(t, n)
    _u8(grid, "grid")
    _i32(rays, "rays")
    _i32(counter, "counter")
    lib = _lib.lib()
    nbytes = int(lib.sf_march_rays_train_workspace_bytes(int(N)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=rays_o.device)       # per-ray counts and scan totals
    _lib.check(lib.sf_march_rays_train(_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(grid), float(bound), float(dt_gamma),
                                       int(max_steps), int(N), int(C), int(H), int(M), _lib.ptr(nears), _lib.ptr(fars),
                                       _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas), _lib.ptr(rays), _lib.ptr(counter),
                                       _lib.ptr(noises), _lib.ptr(ws), nbytes, _lib.stream_ptr()), "march_rays_train")


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    for t, n in ((sigmas, "sigmas"), (rgbs, "rgbs"), (deltas, "deltas"), (weights_sum, "weights_sum"), (depth, "depth"),
                 (image, "image")):
        _f32(t, n)
    _i32(rays, "rays")
    _lib.check(_lib.lib().sf_composite_rays_train_forward(_lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(rays),
                                                          int(M), int(N), float(T_thresh), _lib.ptr(weights_sum),
                                                          _lib.ptr(depth), _lib.ptr(image), _lib.stream_ptr()),
               "composite_rays_train_forward")


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                                  grad_sigmas, grad_rgbs):
    for t, n in ((grad_weights_sum, "grad_weights_sum"), (grad_image, "grad_image"), (sigmas, "sigmas"), (rgbs, "rgbs"),
                 (deltas, "deltas"), (weights_sum, "weights_sum"), (image, "image"), (grad_sigmas, "grad_sigmas"),
                 (grad_rgbs, "grad_rgbs")):
        _f32(t, n)
    _i32(rays, "rays")
    _lib.check(_lib.lib().sf_composite_rays_train_backward(
        _lib.ptr(grad_weights_sum), _lib.ptr(grad_image), _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(rays),
        _lib.ptr(weights_sum), _lib.ptr(image), int(M), int(N), float(T_thresh), _lib.ptr(grad_sigmas), _lib.ptr(grad_rgbs),
        _lib.stream_ptr()), "composite_rays_train_backward")


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs,
               dirs, deltas, noises):
    for t, n in ((rays_t, "rays_t"), (rays_o, "rays_o"), (rays_d, "rays_d"), (nears, "nears"), (fars, "fars"), (xyzs, "xyzs"),
                 (dirs, "dirs"), (deltas, "deltas"), (noises, "noises")):
        _f32(t, n)
    _u8(grid, "grid")
    _i32(rays_alive, "rays_alive")
    _lib.check(_lib.lib().sf_march_rays(int(n_alive), int(n_step), _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(rays_o),
                                        _lib.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C), int(H),
                                        _lib.ptr(grid), _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(xyzs), _lib.ptr(dirs),
                                        _lib.ptr(deltas), _lib.ptr(noises), _lib.stream_ptr()), "march_rays")


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    for t, n in ((rays_t, "rays_t"), (sigmas, "sigmas"), (rgbs, "rgbs"), (deltas, "deltas"), (weights_sum, "weights_sum"),
                 (depth, "depth"), (image, "image")):
        _f32(t, n)
    _i32(rays_alive, "rays_alive")
    _lib.check(_lib.lib().sf_composite_rays(int(n_alive), int(n_step), float(T_thresh), _lib.ptr(rays_alive), _lib.ptr(rays_t),
                                            _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(weights_sum),
                                            _lib.ptr(depth), _lib.ptr(image), _lib.stream_ptr()), "composite_rays")
