"""Volume renderer with the reference's call surface on the fused HIP render.

Mirrors the public behaviour of external/nerf/renderer_df.py:64-717 for the configuration the
distillation loop uses (cuda_ray=False, shading='albedo', bg_radius=0): `render`,
`render_batched`, `run`, buffers `aabb_train` / `aabb_infer`, result dict keys
`image [B,N,3]`, `depth [B,N]`, `weights_sum [N]`, `mask [B,N]`.  The whole of `run`
(:310-468) is ONE autograd node backed by sf_ngp_render_forward / _backward.

RNG: the reference draws, in this order, randn(3) (light direction, unused for 'albedo'),
rand(N,T) (stratified jitter, if perturb) and rand(N,T) (inverse-CDF draw, if training)
(renderer_df.py:351,363,31); `run` draws tensors of the same shapes, roles and order, but on
the DEVICE generator (the reference's `sample_pdf` draws on the CPU generator and copies), and
`update_extra_state` draws its jitter in Morton order: a seeded run is reproducible here, it
does not reproduce the reference's random stream.  Parity tests inject the draws via `noise=`."""
import argparse
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import _lib


def get_default_torch_ngp_opt():
    """Options namespace of sparsefusion/distillation.py:500-525 (same field names)."""
    opt = argparse.Namespace()
    opt.cuda_ray = False
    opt.max_steps = 256
    opt.num_steps = 64
    opt.upsample_steps = 64
    opt.update_extra_interval = 16
    opt.max_ray_batch = 4096
    opt.albedo_iters = 1000
    opt.bg_radius = 0
    opt.density_thresh = 10
    opt.fp16 = True
    opt.backbone = 'grid'
    opt.w = 128
    opt.h = 128
    opt.hw_scale = 2
    opt.bound = 4
    opt.min_near = 0.1
    opt.dt_gamma = 0
    opt.lambda_entropy = 1e-4
    opt.lambda_opacity = 0
    opt.lambda_orient = 1e-2
    opt.lambda_smooth = 0
    return opt


class _FieldHandle:
    """ctypes view of the field parameters; keeps the host offsets alive."""

    def __init__(self, net):
        enc = net.encoder
        lin = net.sigma_net.net
        self.tensors = [enc.embeddings, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight,
                        lin[2].bias]
        self.host_offsets = enc.host_offsets
        self.L = enc.num_levels
        self.S = float(math.log2(enc.per_level_scale))
        self.H = int(enc.base_resolution)
        self.gridtype = int(enc.gridtype_id)
        self.bound = float(net.bound)

    def struct(self, tensors):
        f = _lib.SfNgpField()
        f.embeddings = tensors[0].data_ptr()
        f.h_offsets = self.host_offsets.ctypes.data
        f.L, f.S, f.H, f.gridtype = self.L, self.S, self.H, self.gridtype
        f.w0, f.b0, f.w1, f.b1, f.w2, f.b2 = (t.data_ptr() for t in tensors[1:])
        f.bound = self.bound
        return f


_FEAT_CACHE = True        # False: the backward re-gathers the features (the C ABI's field_cache = NULL path; tests flip this module attribute)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, rays_o, rays_d, aabb, T, min_near, lin, u_coarse, u_fine, u_stride, bg, rays_per_row, grad_mode, *params):
        _lib.require_cuda(rays_o, rays_d, aabb, *params)
        params = [p.detach().contiguous() for p in params]
        N = rays_o.shape[0]
        dev = rays_o.device
        f32 = dict(dtype=torch.float32, device=dev)
        nears, fars = torch.empty(N, **f32), torch.empty(N, **f32)
        z_s, sig_s = torch.empty(N, 2 * T, **f32), torch.empty(N, 2 * T, **f32)
        rgb_s = torch.empty(N, 2 * T, 3, **f32)
        image, depth, ws = torch.empty(N, 3, **f32), torch.empty(N, **f32), torch.empty(N, **f32)
        lib = _lib.lib()
        wbytes = lib.sf_ngp_render_forward_workspace_bytes(N, T)
        work = torch.empty(max(1, wbytes // 4), **f32)
        f = handle.struct(params)
        # field cache (r03): when a backward will follow, the forward keeps the hash-grid features of every sample and the sort
        # permutation, so the backward reads 128 bytes per sample instead of re-gathering 16 levels x 8 corners (0.73 ms of a
        # 4.5 ms backward at 128^2 rays x 128 samples; 268 MB per render held until then).  `_FEAT_CACHE = False`: recompute.
        cache = None
        # `grad_mode` = torch.is_grad_enabled() at the call site: needs_input_grad mirrors requires_grad even under no_grad(), and
        # grad mode is always off in here -- an eval render of a trainable field must not allocate / write the 268 MB cache
        if _FEAT_CACHE and grad_mode and any(ctx.needs_input_grad[13:]):
            cache = torch.empty(lib.sf_ngp_render_cache_bytes(N, T) // 4, **f32)
        rc = lib.sf_ngp_render_forward(C.byref(f), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), N, T,
                                       float(min_near), _lib.ptr(lin), _lib.ptr(u_coarse), _lib.ptr(u_fine),
                                       int(u_stride), float(bg), _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(z_s),
                                       _lib.ptr(sig_s), _lib.ptr(rgb_s), _lib.ptr(image), _lib.ptr(depth), _lib.ptr(ws),
                                       _lib.ptr(cache), _lib.ptr(work), wbytes, _lib.stream_ptr())
        _lib.check(rc, "ngp_render_forward")
        ctx.handle, ctx.T, ctx.bg, ctx.rays_per_row = handle, T, float(bg), int(rays_per_row)
        ctx.has_cache = cache is not None
        ctx.save_for_backward(rays_o, rays_d, aabb, nears, fars, z_s, sig_s, rgb_s, *([cache] if cache is not None else []), *params)
        ctx.mark_non_differentiable(depth, nears, fars)
        return image, ws, depth, nears, fars

    @staticmethod
    def backward(ctx, g_image, g_ws, _gd, _gn, _gf):
        rays_o, rays_d, aabb, nears, fars, z_s, sig_s, rgb_s, *params = ctx.saved_tensors
        cache = params.pop(0) if ctx.has_cache else None
        N, T = rays_o.shape[0], ctx.T
        grads = [torch.zeros_like(p) for p in params]
        g = _lib.SfNgpFieldGrad()
        (g.g_embeddings, g.g_w0, g.g_b0, g.g_w1, g.g_b1, g.g_w2, g.g_b2) = (t.data_ptr() for t in grads)
        f = ctx.handle.struct(params)
        lib = _lib.lib()
        wbytes = lib.sf_ngp_render_workspace_bytes(N, T)
        work = torch.empty(max(1, wbytes // 4), dtype=torch.float32, device=rays_o.device)
        g_image = (g_image if g_image is not None else torch.zeros(N, 3, device=rays_o.device)).contiguous().float()
        g_ws = g_ws.contiguous().float() if g_ws is not None else None
        rc = lib.sf_ngp_render_backward(C.byref(f), C.byref(g), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), N, T,
                                        _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(z_s), _lib.ptr(sig_s), _lib.ptr(rgb_s),
                                        ctx.bg, _lib.ptr(g_image), _lib.ptr(g_ws), int(ctx.rays_per_row), _lib.ptr(cache),
                                        _lib.ptr(work), wbytes, _lib.stream_ptr())
        _lib.check(rc, "ngp_render_backward")
        return (None,) * 13 + tuple(grads)


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.bound = opt.bound
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.grid_size = 128
        self.cuda_ray = opt.cuda_ray
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh
        self.bg_radius = opt.bg_radius
        if self.bg_radius > 0:
            raise NotImplementedError("bg_radius > 0 is not on the distillation path (distillation.py:512)")
        box = torch.tensor([-opt.bound] * 3 + [opt.bound] * 3, dtype=torch.float32)
        self.register_buffer('aabb_train', box)
        self.register_buffer('aabb_infer', box.clone())
        self._tables = {}
        if self.cuda_ray:                                        # extra state of the occupancy-grid path (renderer_df.py:85-97)
            self.register_buffer('density_grid', torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer('density_bitfield', torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    # ---- hooks implemented by the field (network_grid.NeRFNetwork)
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        """renderer_df.py:108-119."""
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, noise=None):
        """EMA update of the cascaded density grid, its bitfield and the mean sample count (renderer_df.py:586-638).

        The grid is stored in Morton order, so every cell of a cascade is visited in storage order: cell coordinates
        come from ONE morton3D_invert of arange(H^3) and the queried densities are written back contiguously (the
        reference walks the grid in x-major blocks of S^3 and scatters through morton3D; same cells, same jitter law).
        `noise(like) -> U[0,1)` injects the per-cascade jitter in the REFERENCE's x-major cell order (tests); by default it
        is drawn directly in storage order."""
        if not self.cuda_ray:
            return
        from .. import raymarching
        H, dev = self.grid_size, self.density_bitfield.device
        coords = raymarching.morton3D_invert(torch.arange(H ** 3, dtype=torch.int32, device=dev))      # [H^3, 3] of cell (x, y, z)
        xyzs = 2 * coords.float() / (H - 1) - 1
        xmajor = (coords[:, 0].long() * H + coords[:, 1].long()) * H + coords[:, 2].long() if noise is not None else None
        tmp_grid = torch.empty_like(self.density_grid)
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            half_cell = bound / H
            u = noise(xyzs)[xmajor] if noise is not None else torch.rand_like(xyzs)
            pts = xyzs * (bound - half_cell) + (u * 2 - 1) * half_cell
            tmp_grid[cas] = self.density(pts)['sigma'].reshape(-1)
        valid = self.density_grid >= 0                                    # cells never marked invalid (< 0) take the EMA / max rule
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp_grid[valid])
        self.mean_density = torch.mean(self.density_grid[valid]).item()
        self.iter_density += 1
        self.density_bitfield = raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh),
                                                     self.density_bitfield)
        rounds = min(16, self.local_step)                                 # point budget of the next rounds = mean of the last ones
        if rounds > 0:
            self.mean_count = int(self.step_counter[:rounds, 0].sum().item() / rounds)
        self.local_step = 0

    def _table(self, T, device):
        key = (T, str(device))
        if key not in self._tables:
            # built on the HOST and copied (once per (T, device)): torch's device linspace kernel and its CPU kernel round some
            # entries differently, and the oracle -- bit-identical to the reference's own `run` on CPU -- is the parity anchor of the
            # coarse sample depths (tests/test_gpu_ngp.py::test_render_sample_bookkeeping_vs_oracle)
            lin = torch.linspace(0.0, 1.0, T).to(device)
            det = torch.linspace(0. + 0.5 / T, 1. - 0.5 / T, steps=T).to(device)
            self._tables[key] = (lin.contiguous(), det.contiguous())
        return self._tables[key]

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, light_d=None, ambient_ratio=1.0,
            shading='albedo', bg_color=None, perturb=False, fixed_light=False, noise=None, **kwargs):
        """rays_o, rays_d: [B, N, 3] (B == 1).  `noise` = dict(u_coarse=[N,T], u_fine=[N,T]) injects the draws."""
        if shading != 'albedo':
            raise NotImplementedError("only shading='albedo' is on the distillation path (distillation.py:209,282)")
        if num_steps != upsample_steps:
            raise NotImplementedError("the fused render needs num_steps == upsample_steps (reference: 64/64)")
        prefix = rays_o.shape[:-1]
        o = rays_o.contiguous().view(-1, 3).float()
        d = rays_d.contiguous().view(-1, 3).float()
        N, T, dev = o.shape[0], int(num_steps), o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        lin, det = self._table(T, dev)
        if noise is None:
            if light_d is None and not fixed_light:
                torch.randn(3, device=dev, dtype=torch.float)          # consumed, unused for 'albedo' (:351)
            u_coarse = torch.rand(N, T, device=dev) if perturb else None   # :363
            u_fine = torch.rand(N, T, device=dev) if self.training else None  # sample_pdf det=not training (:31)
        else:
            u_coarse, u_fine = noise.get("u_coarse"), noise.get("u_fine")
        u_f, stride = (u_fine.contiguous(), T) if u_fine is not None else (det, 0)
        if u_coarse is not None:
            u_coarse = u_coarse.contiguous()
        if bg_color is None:
            bg_color = 1
        if torch.is_tensor(bg_color):
            raise NotImplementedError("per-ray bg_color tensors are not on the distillation path (bg_color=0)")
        image, weights_sum, depth, nears, fars = _RenderFn.apply(
            self._field_handle(), o, d, aabb, T, self.min_near, lin, u_coarse, u_f, stride, float(bg_color),
            self._rays_per_row(N, kwargs), torch.is_grad_enabled(), *self._field_params())
        return {'image': image.view(*prefix, 3), 'depth': depth.view(*prefix), 'weights_sum': weights_sum,
                'mask': (nears < fars).view(*prefix)}

    @staticmethod
    def _rays_per_row(N, kwargs):
        """Image width when the rays are a full row-major w x h image (opt.w / opt.h travel in **vars(opt)): a
        locality hint for the gradient scatter only, results do not depend on it."""
        w, h = kwargs.get('w'), kwargs.get('h')
        return int(w) if w and h and int(w) * int(h) == N else 0

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, light_d=None, ambient_ratio=1.0, shading='albedo', bg_color=None,
                 perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, noise=None, **kwargs):
        """Render through the occupancy grid -- the call surface and results of renderer_df.py:471-584 (`cuda_ray=True`),
        restructured for the GPU:
          * evaluation is ONE launch (sf_ngp_render_occ_eval): every lane walks its ray through the density bitfield and
            evaluates the fused field at the occupied samples until it is opaque / leaves the box.  The reference's rounds of
            march_rays -> network -> composite_rays over a shrinking alive list need a host read per round; with
            perturb=False (the reference's evaluation setting, renderer_df.py:653-717 via render_batched) the samples of a
            ray and the arithmetic on them do not depend on that batching, so the result is the same.  DEVIATION with
            perturb=True: the reference restarts every round from rays_t = near + sum of the deltas it composited, which
            lags the jittered t by step * noise (raymarching.cu:736-748, renderer_df.py:548), so later rounds re-jitter
            from a slightly earlier t; this kernel keeps marching from the true t.  The sample sets differ by less than
            one step per round; also the unused `light_d` randn(3) draw of the reference is not consumed (RNG stream
            differs).  Use the raymarching.march_rays / composite_rays entry points for the round-exact behaviour;
          * training keeps the three stages (the samples must exist as tensors for autograd): slot assignment by a block
            scan (raymarching.march_rays_train), field query through the differentiable grid-encode op, compositing with
            its own backward kernel.
        `noise` = per-ray jitter tensor [N] in place of torch.rand."""
        from .. import raymarching
        if shading != 'albedo':
            raise NotImplementedError("only shading='albedo' is on the distillation path (distillation.py:209,282)")
        lead = rays_o.shape[:-1]
        o = rays_o.contiguous().view(-1, 3).float()
        d = rays_d.contiguous().view(-1, 3).float()
        box = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(o, d, box)
        stage = self._occ_train if self.training else self._occ_eval
        opacity, z, rgb = stage(o, d, nears, fars, dt_gamma, perturb, force_all_rays, max_steps, T_thresh, noise, ambient_ratio)
        background = 1 if bg_color is None else bg_color
        rgb = rgb + (1 - opacity).unsqueeze(-1) * background
        return {'image': rgb.view(*lead, 3),
                'depth': (torch.clamp(z - nears, min=0) / (fars - nears)).view(*lead),
                'weights_sum': opacity.reshape(*lead),
                'mask': (nears < fars).reshape(*lead)}

    def _occ_train(self, o, d, nears, fars, dt_gamma, perturb, force_all_rays, max_steps, T_thresh, noise, ambient_ratio):
        from .. import raymarching
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                                nears, fars, counter, self.mean_count, perturb, 128, force_all_rays,
                                                                dt_gamma, max_steps, noises=noise)
        sigmas, rgbs, _ = self(xyzs, dirs, None, ratio=ambient_ratio, shading='albedo')
        opacity, z, rgb = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
        return opacity, z, rgb

    @torch.no_grad()
    def _occ_eval(self, o, d, nears, fars, dt_gamma, perturb, force_all_rays, max_steps, T_thresh, noise, ambient_ratio):
        N, dev = o.shape[0], o.device
        if perturb and noise is None:
            noise = torch.rand(N, dtype=torch.float32, device=dev)
        jitter = noise.float().contiguous() if (perturb and noise is not None) else None
        opacity = torch.empty(N, dtype=torch.float32, device=dev)
        z = torch.empty(N, dtype=torch.float32, device=dev)
        rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        params = [p.detach().contiguous() for p in self._field_params()]
        f = self._field_handle().struct(params)
        rc = _lib.lib().sf_ngp_render_occ_eval(C.byref(f), _lib.ptr(o), _lib.ptr(d), _lib.ptr(nears), _lib.ptr(fars),
                                               _lib.ptr(self.density_bitfield.contiguous()), float(dt_gamma), int(max_steps),
                                               int(self.cascade), int(self.grid_size), _lib.ptr(jitter), float(T_thresh), N,
                                               _lib.ptr(opacity), _lib.ptr(z), _lib.ptr(rgb), _lib.stream_ptr())
        _lib.check(rc, "ngp_render_occ_eval")
        return opacity, z, rgb

    def _run(self, rays_o, rays_d, **kwargs):
        """renderer_df.py:647-650: the occupancy-grid marcher when cuda_ray, the fused coarse+fine sampler otherwise."""
        return (self.run_cuda if self.cuda_ray else self.run)(rays_o, rays_d, **kwargs)

    def _chunked(self, rays_o, rays_d, max_ray_batch, kwargs):
        B, N = rays_o.shape[:2]
        dev = rays_o.device
        depth, image = torch.empty((B, N), device=dev), torch.empty((B, N, 3), device=dev)
        weights_sum = torch.empty((B, N), device=dev)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                r = self._run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = r['depth']
                weights_sum[b:b + 1, head:tail] = r['weights_sum']
                image[b:b + 1, head:tail] = r['image']
        return {'depth': depth, 'image': image, 'weights_sum': weights_sum}

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """renderer_df.py:643-679: one `run` over all rays, or chunks of max_ray_batch when staged."""
        if staged and not self.cuda_ray:                         # "never stage when cuda_ray" (:655)
            return self._chunked(rays_o, rays_d, max_ray_batch, kwargs)
        return self._run(rays_o, rays_d, **kwargs)

    def render_batched(self, rays_o, rays_d, batched=False, max_ray_batch=128 * 128, **kwargs):
        """renderer_df.py:681-717: no-grad render, optionally in chunks of max_ray_batch rays."""
        kwargs.pop('max_ray_batch', None)
        with torch.no_grad():
            if batched:
                return self._chunked(rays_o, rays_d, max_ray_batch, kwargs)
            return self._run(rays_o, rays_d, **kwargs)
