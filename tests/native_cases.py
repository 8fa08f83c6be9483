"""Seeded inputs for the natives of the NGP path (grid encode fwd / bwd, near-far, sph, morton, packbits, occupancy marching
and compositing) and one driver per family that runs them through ANY binding with the reference's positional signatures
(oracle.ngp_native = the C restatement, oracle.ref_native = the reference's own .cu files compiled for the host).
Used by tests/golden/make_golden_native.py (writes tests/golden/ngp_native.pt) and tests/test_oracle_native_pin.py."""
import hashlib
import itertools

import numpy as np
import torch

import occ_common as oc

GRID_CONFIGS = list(itertools.product((0, 1), (1, 2, 3), (1, 2, 4, 8), (False, True)))     # gridtype, D, C, align_corners
MARCH_CONFIGS = [(0.0, True), (1.0 / 128, True), (0.0, False)]                                # dt_gamma, perturb


def digest(t):
    """sha256 over dtype, shape and the raw bytes (NaN payloads included): equality of digests = bit equality."""
    t = t.detach().cpu().contiguous()
    h = hashlib.sha256(f"{t.dtype}{tuple(t.shape)}".encode())
    h.update(t.numpy().tobytes())
    return h.hexdigest()


def grid_case(gridtype, D, C, align_corners, seed=0, L=6, H=4, B=96, scale=1.7, cap=2 ** 9):
    """Levels 0..2 are dense at D = 3 (tiled rows), the rest hashed / wrapped; rows 0-3 of the inputs sit on the domain
    boundary (0, 1) and outside it (the kernels return zeros there: gridencoder.cu:107-121)."""
    g = torch.Generator().manual_seed(seed * 1000 + gridtype * 100 + D * 10 + C + (5 if align_corners else 0))
    S = float(np.log2(scale))
    offs = [0]
    for l in range(L):
        res = int(np.ceil(H * np.exp2(l * S)))
        p = min(cap, (res if align_corners else res + 1) ** D)
        offs.append(offs[-1] + int(np.ceil(p / 8) * 8))
    x = torch.rand(B, D, generator=g)
    x[0], x[1], x[2, 0], x[3, 0] = 0.0, 1.0, -0.1, 1.5
    return dict(x=x, emb=torch.randn(offs[-1], C, generator=g) * 0.1, offsets=torch.tensor(offs, dtype=torch.int32),
                g=torch.randn(L, B, C, generator=g), B=B, D=D, C=C, L=L, S=S, H=H, gridtype=gridtype, ac=align_corners)


def run_grid(mod, c):
    out = torch.zeros(c["L"], c["B"], c["C"])
    dy_dx = torch.zeros(c["B"], c["L"] * c["D"] * c["C"])
    mod.grid_encode_forward(c["x"], c["emb"], c["offsets"], out, c["B"], c["D"], c["C"], c["L"], c["S"], c["H"], dy_dx,
                            c["gridtype"], c["ac"])
    ge, gi = torch.zeros_like(c["emb"]), torch.zeros(c["B"], c["D"])
    mod.grid_encode_backward(c["g"], c["x"], c["emb"], c["offsets"], ge, c["B"], c["D"], c["C"], c["L"], c["S"], c["H"], dy_dx,
                             gi, c["gridtype"], c["ac"])
    return dict(outputs=out, dy_dx=dy_dx, grad_embeddings=ge, grad_inputs=gi)


def run_raymarching(mod, dt_gamma=0.0, perturb=True, seed=0, n_side=10):
    """Every raymarching entry point on one scene (tests/occ_common.py: a ball in bound 4, 3 cascades): utilities on random
    rays incl. axis-parallel directions and a ray that misses the box; training march + composite fwd / bwd; three rounds
    of the inference loop (march_rays -> composite_rays -> compaction of the alive list as renderer_df.py:520-560 does)."""
    out = {}
    g = torch.Generator().manual_seed(seed)
    N = 300
    o = (torch.rand(N, 3, generator=g) - 0.5) * 6
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    d[0], d[1] = torch.tensor([0., 0., 1.]), torch.tensor([1., 0., 0.])                     # zero direction components
    o[2], d[2] = torch.tensor([9., 9., 9.]), torch.tensor([1., 0., 0.])                     # misses the box
    aabb = torch.tensor([-4., -4, -4, 4, 4, 4])
    nears, fars = torch.empty(N), torch.empty(N)
    mod.near_far_from_aabb(o, d, aabb, N, 0.05, nears, fars)
    out["nears"], out["fars"] = nears, fars
    sph = torch.empty(N, 2)
    mod.sph_from_ray(o, d, 5.0, N, sph)
    out["sph"] = sph
    co = torch.randint(0, 128, (1000, 3), generator=g, dtype=torch.int32)
    ind = torch.empty(1000, dtype=torch.int32)
    mod.morton3D(co, 1000, ind)
    co2 = torch.empty(1000, 3, dtype=torch.int32)
    mod.morton3D_invert(ind, 1000, co2)
    out["morton"], out["morton_invert"] = ind, co2
    gridf = torch.rand(8 * 500, generator=g)
    bits = torch.empty(500, dtype=torch.uint8)
    mod.packbits(gridf, 500, 0.5, bits)
    out["packbits"] = bits
    # training march + composite
    _, bf, _ = oc.ball_bitfield()
    ro, rd = oc.camera_rays(n_side)
    N = ro.shape[0]
    aabb = torch.tensor([-oc.BOUND] * 3 + [oc.BOUND] * 3)
    nn, ff = torch.empty(N), torch.empty(N)
    mod.near_far_from_aabb(ro, rd, aabb, N, 0.05, nn, ff)
    noises = torch.rand(N, generator=g) if perturb else torch.zeros(N)
    M = N * 256
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays = torch.full((N, 3), -7, dtype=torch.int32)
    counter = torch.zeros(2, dtype=torch.int32)
    mod.march_rays_train(ro, rd, bf, oc.BOUND, dt_gamma, oc.MAX_STEPS, N, oc.CASCADE, oc.H, M, nn, ff, xyzs, dirs, deltas, rays,
                         counter, noises)
    Mu = int(counter[0])
    out.update(march_xyzs=xyzs[:Mu], march_dirs=dirs[:Mu], march_deltas=deltas[:Mu], march_rays=rays, march_counter=counter)
    sig, rgb = torch.rand(M, generator=g) * 3, torch.rand(M, 3, generator=g)
    ws, dep, img = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    mod.composite_rays_train_forward(sig, rgb, deltas, rays, Mu, N, 1e-4, ws, dep, img)
    out.update(comp_weights_sum=ws, comp_depth=dep, comp_image=img)
    gws, gim = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    gs, gr = torch.zeros(M), torch.zeros(M, 3)
    mod.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws, img, Mu, N, 1e-4, gs, gr)
    out.update(comp_grad_sigmas=gs[:Mu], comp_grad_rgbs=gr[:Mu])
    # inference rounds
    n_alive, alive, rt = N, torch.arange(N, dtype=torch.int32), nn.clone()
    ws2, dep2, img2 = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    for rnd in range(3):
        n_step = 8
        x2, d2, dl2 = torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 2)
        mod.march_rays(n_alive, n_step, alive, rt, ro, rd, oc.BOUND, dt_gamma, oc.MAX_STEPS, oc.CASCADE, oc.H, bf, nn, ff, x2, d2,
                       dl2, noises)
        sg = ((x2.norm(dim=-1) < 1.5).float() * 2.0).contiguous()
        cl = torch.sigmoid(x2).contiguous()
        mod.composite_rays(n_alive, n_step, 1e-2, alive, rt, sg, cl, dl2, ws2, dep2, img2)
        out[f"inf{rnd}_xyzs"], out[f"inf{rnd}_deltas"] = x2, dl2
        out[f"inf{rnd}_alive"], out[f"inf{rnd}_t"] = alive.clone(), rt.clone()
        alive = alive[:n_alive][alive[:n_alive] >= 0].contiguous()
        n_alive = alive.numel()
        if n_alive == 0:
            break
    out.update(inf_weights_sum=ws2, inf_depth=dep2, inf_image=img2)
    return out


# outputs of the contracted ("fused", nvcc -fmad=true analogue) build that are NOT expected bit-equal between g++'s and the
# oracle's contraction choices (sums of products: `ws += weight`, `r += weight * rgb`, __expf): compared with a tolerance
FUSED_TOLERANT = ("sph", "comp_weights_sum", "comp_depth", "comp_image", "comp_grad_sigmas", "comp_grad_rgbs",
                  "inf_weights_sum", "inf_depth", "inf_image")
# the inference loop's alive list / t depend on `T < T_thresh` of a 1-ulp-different T: kept out of the bit-exact set too
FUSED_DEPENDENT = tuple(f"inf{r}_{k}" for r in (1, 2) for k in ("xyzs", "deltas", "alive", "t")) + ("inf0_alive", "inf0_t")
