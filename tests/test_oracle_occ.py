"""First-principles checks of the occupancy-grid oracle (oracle/ngp_ref.c, second half).  The reference has no
test, golden or CPU path for these CUDA kernels (parity unpinned); what CAN be pinned is what
the algorithm promises: samples only in occupied cells, contiguous per-ray slots, telescoping deltas, the
volume-rendering sums and their analytic gradient, train / inference consistency."""
import math

import pytest
import torch

from oracle import ngp_native
from occ_common import BOUND, CASCADE, H, MAX_STEPS, ball_bitfield, camera_rays, near_far, oracle_march_train


@pytest.fixture(scope="module")
def scene():
    grid, bits, center = ball_bitfield()
    o, d = camera_rays()
    nears, fars = near_far(o, d)
    return grid, bits, center, o, d, nears, fars


@pytest.mark.parametrize("dt_gamma,perturb", [(0.0, False), (0.0, True), (1.0 / 128, True)])
def test_march_rays_train_geometry(scene, dt_gamma, perturb):
    grid, bits, center, o, d, nears, fars = scene
    N = o.shape[0]
    noises = torch.rand(N, generator=torch.Generator().manual_seed(1)) if perturb else torch.zeros(N)
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, noises, dt_gamma)
    counts = rays[:, 2].long()
    assert rays[:, 0].tolist() == list(range(N))                                # serial schedule: ray order
    assert torch.equal(rays[:, 1].long(), torch.cumsum(counts, 0) - counts)     # contiguous slots
    assert counter.tolist() == [int(counts.sum()), N]
    total = int(counts.sum())
    assert total > 2000 and (counts == 0).any() and (counts > 0).any()          # some rays miss the ball
    pts = xyzs[:total]
    assert (xyzs[total:] == 0).all()
    # every sample sits in a cell whose centre is inside the ball -> within half a cell diagonal of the ball
    level = torch.maximum(torch.frexp(pts.abs().amax(-1))[1], torch.frexp(deltas[:total, 0] * H * 0.5)[1])   # cascade picked by
    level = level.clamp(0, CASCADE - 1).float()                                                    # position AND step size
    cell = 2 * torch.minimum(2.0 ** level, torch.tensor(BOUND)) / H
    dist = (pts - center).norm(dim=-1)
    assert (dist <= 1.3 + cell * math.sqrt(3) / 2 * 1.001 + 1e-5).all()
    # and the ball is actually sampled: rays through the centre region collect ~ chord / dt samples
    dt_min = 2 * math.sqrt(3) / MAX_STEPS
    assert counts.max() <= MAX_STEPS
    if dt_gamma == 0:
        assert torch.allclose(deltas[:total, 0], torch.full((total,), dt_min))
        chord_steps = 2 * 1.3 / (dt_min * d.norm(dim=-1).max())                 # |d| != 1: t is in units of |d|
        assert counts.max() <= chord_steps * 1.15 + 3 and counts.max() >= chord_steps * 0.5
    # positions are o + t d with t advancing by deltas[:, 1] (the first one also spans the empty space from `near`)
    for n in (int(torch.argmax(counts)), int((counts > 0).nonzero()[0])):
        off, cnt = int(rays[n, 1]), int(rays[n, 2])
        dl = deltas[off:off + cnt]
        assert (dl[:, 1] >= dl[:, 0] - 2e-6).all()                              # gaps (skipped empty cells) only add
        t_end = torch.cumsum(dl[:, 1].double(), 0)
        t_start = t_end - dl[:, 0].double()
        p = xyzs[off:off + cnt].double()
        # solve t from the dominant axis and compare increments with the recorded deltas
        ax = int(d[n].abs().argmax())
        t = (p[:, ax] - o[n, ax].double()) / d[n, ax].double()
        assert torch.allclose(t - t[0], t_start - t_start[0], atol=2e-5)
        assert torch.allclose(dirs[off:off + cnt], d[n].expand(cnt, 3))


def test_march_rays_train_overflow_and_counter_base(scene):
    """M too small: rays whose slots do not fit are recorded but not written (raymarching.cu:415); the counter keeps
    counting, and a non-zero incoming counter offsets the slots as atomicAdd's return value does."""
    grid, bits, center, o, d, nears, fars = scene
    N = o.shape[0]
    noises = torch.zeros(N)
    full = oracle_march_train(o, d, bits, nears, fars, noises)
    total = int(full[4][0])
    M = total // 2
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, noises, M=M)
    assert counter.tolist() == [total, N] and torch.equal(rays, full[3])
    fits = (rays[:, 1] + rays[:, 2] <= M)
    last = int((rays[fits, 1] + rays[fits, 2]).max())
    assert torch.equal(xyzs[:last], full[0][:last]) and (xyzs[last:] == 0).all() and last <= M
    # restart from a non-zero point counter (ray counter must start at 0: rays has exactly N rows)
    base = torch.tensor([100, 0], dtype=torch.int32)
    x2, _, _, r2, c2 = oracle_march_train(o, d, bits, nears, fars, noises, M=total + 100, counter=base)
    assert c2.tolist() == [total + 100, N] and torch.equal(r2[:, 1], full[3][:, 1] + 100)
    assert torch.equal(x2[100:100 + total], full[0][:total])


def _torch_composite(sigmas, rgbs, deltas, rays, N):
    ws, depth, image = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    ws_l, dp_l, im_l = [], [], []
    for n in range(rays.shape[0]):
        idx, off, cnt = (int(v) for v in rays[n])
        if cnt == 0:
            ws_l.append(torch.zeros(())); dp_l.append(torch.zeros(())); im_l.append(torch.zeros(3)); continue
        s, c, dl = sigmas[off:off + cnt], rgbs[off:off + cnt], deltas[off:off + cnt]
        alpha = 1 - torch.exp(-s * dl[:, 0])
        T = torch.cumprod(torch.cat([torch.ones(1), 1 - alpha]), 0)[:-1]
        w = alpha * T
        t = torch.cumsum(dl[:, 1], 0)
        ws_l.append(w.sum()); dp_l.append((w * t).sum()); im_l.append((w[:, None] * c).sum(0))
    return torch.stack(ws_l), torch.stack(dp_l), torch.stack(im_l)


def test_composite_train_forward_backward_against_autograd(scene):
    grid, bits, center, o, d, nears, fars = scene
    N = o.shape[0]
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, torch.zeros(N))
    M = int(counter[0])
    g = torch.Generator().manual_seed(3)
    sigmas = (torch.rand(M, generator=g) * 3).requires_grad_(True)
    rgbs = torch.rand(M, 3, generator=g).requires_grad_(True)
    deltas = deltas[:M].contiguous()
    ws, depth, image = torch.empty(N), torch.empty(N), torch.empty(N, 3)
    ngp_native.composite_rays_train_forward(sigmas.detach(), rgbs.detach(), deltas, rays, M, N, 0.0, ws, depth, image)
    ws_r, dp_r, im_r = _torch_composite(sigmas, rgbs, deltas, rays, N)
    assert torch.allclose(ws, ws_r.detach(), atol=2e-6) and torch.allclose(image, im_r.detach(), atol=2e-6)
    assert torch.allclose(depth, dp_r.detach(), rtol=1e-5, atol=1e-5)
    gws, gim = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    (ws_r * gws).sum().backward(retain_graph=True)
    (im_r * gim).sum().backward()
    gs, gc = torch.zeros(M), torch.zeros(M, 3)
    ngp_native.composite_rays_train_backward(gws, gim, sigmas.detach(), rgbs.detach(), deltas, rays, ws, image, M, N, 0.0, gs, gc)
    assert torch.allclose(gc, rgbs.grad, atol=2e-6)
    assert torch.allclose(gs, sigmas.grad, rtol=2e-4, atol=2e-5)
    # early termination: once T < T_thresh the remaining samples contribute nothing and get zero gradient
    ws2, dp2, im2 = torch.empty(N), torch.empty(N), torch.empty(N, 3)
    big = sigmas.detach() * 40
    ngp_native.composite_rays_train_forward(big, rgbs.detach(), deltas, rays, M, N, 1e-4, ws2, dp2, im2)
    assert (ws2 <= 1 + 1e-6).all() and (ws2[rays[:, 2] > 200] > 0.999).all()
    gs2, gc2 = torch.zeros(M), torch.zeros(M, 3)
    ngp_native.composite_rays_train_backward(gws, gim, big, rgbs.detach(), deltas, rays, ws2, im2, M, N, 1e-4, gs2, gc2)
    n = int(torch.argmax(rays[:, 2]))
    off, cnt = int(rays[n, 1]), int(rays[n, 2])
    assert (gc2[off + cnt - 5:off + cnt] == 0).all() and (gc2[off] != 0).any()


def test_inference_loop_matches_train_composite(scene):
    """renderer_df.py:523-556: march n_step samples for the alive rays, composite in place, drop finished rays."""
    grid, bits, center, o, d, nears, fars = scene
    N = o.shape[0]
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, torch.zeros(N))
    M = int(counter[0])

    def field(x):                                             # any deterministic function of position
        return (2.0 + torch.sin(3 * x).sum(-1)).clamp(min=0), torch.sigmoid(x * 2)

    sig, col = field(xyzs[:M])
    ws_t, dp_t, im_t = torch.empty(N), torch.empty(N), torch.empty(N, 3)
    ngp_native.composite_rays_train_forward(sig.contiguous(), col.contiguous(), deltas[:M].contiguous(), rays, M, N, 0.0,
                                            ws_t, dp_t, im_t)
    ws, depth, image = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    rays_alive = torch.arange(N, dtype=torch.int32)
    rays_t = nears.clone()
    step, rounds = 0, 0
    while step < MAX_STEPS and rays_alive.numel() > 0:
        n_alive = rays_alive.numel()
        n_step = max(min(N // n_alive, 8), 1)
        x, dr, dl = torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 2)
        ngp_native.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, BOUND, 0.0, MAX_STEPS, CASCADE, H, bits, nears, fars,
                              x, dr, dl, torch.zeros(n_alive))
        s, c = field(x)
        ngp_native.composite_rays(n_alive, n_step, 0.0, rays_alive, rays_t, s.contiguous(), c.contiguous(), dl, ws, depth, image)
        rays_alive = rays_alive[rays_alive >= 0].contiguous()
        step += n_step
        rounds += 1
    assert rounds > 10 and rays_alive.numel() == 0
    assert torch.allclose(ws, ws_t, atol=1e-5) and torch.allclose(image, im_t, atol=1e-5)
    hit = rays[:, 2] > 0
    # inference depth accumulates t from `near`, training from 0: they differ by near * weights_sum (renderer_df.py:574)
    assert torch.allclose((depth - nears * ws)[hit], dp_t[hit], rtol=1e-4, atol=1e-4)


def test_sph_from_ray():
    g = torch.Generator().manual_seed(0)
    o = torch.randn(64, 3, generator=g) * 0.5
    d = torch.randn(64, 3, generator=g)
    coords = torch.empty(64, 2)
    ngp_native.sph_from_ray(o, d, 5.0, 64, coords)
    # independent: intersect |o + t d| = R in double, convert to angles
    od, dd = o.double(), d.double()
    A, B, Cq = (dd * dd).sum(-1), (od * dd).sum(-1), (od * od).sum(-1) - 25.0
    t = (-B + torch.sqrt(B * B - A * Cq)) / A
    p = od + t[:, None] * dd
    assert torch.allclose(p.norm(dim=-1), torch.full((64,), 5.0, dtype=torch.float64), atol=1e-9)
    theta = torch.atan2(torch.sqrt(p[:, 0] ** 2 + p[:, 2] ** 2), p[:, 1])
    phi = torch.atan2(p[:, 2], p[:, 0])
    assert torch.allclose(coords[:, 0].double(), 2 * theta / math.pi - 1, atol=1e-5)
    assert torch.allclose(coords[:, 1].double(), phi / math.pi, atol=1e-5)
