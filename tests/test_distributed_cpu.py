"""World-size-2 tests of the view-sharding helpers on the gloo backend (runs on CPU; RCCL uses the same code)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparsefusion_amd import distributed as sfd
    try:
        # view sharding: 7 views over 2 ranks -> 4 + 3, disjoint, ordered
        mine = sfd.shard_views(7, rank, world)
        assert mine == ([0, 1, 2, 3] if rank == 0 else [4, 5, 6])
        # all-gather of per-rank latents keeps rank order
        lat = torch.full((2, 4, 32, 32), float(rank + 1))
        full = sfd.all_gather_latents(lat)
        assert full.shape == (4, 4, 32, 32) and torch.equal(full[:2], torch.ones(2, 4, 32, 32)) \
            and torch.equal(full[2:], 2 * torch.ones(2, 4, 32, 32))
        # replicas: identical start, different local grads, identical after the mean all-reduce + step
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
        sfd.broadcast_params(net)
        ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
        ref.load_state_dict(net.state_dict())
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        g = torch.Generator().manual_seed(100)
        xs = torch.randn(2, 5, 8, generator=g)                  # the two ranks' "views"
        net(xs[rank]).pow(2).mean().backward()
        sfd.all_reduce_grads(net.parameters())
        opt.step()
        # single-process reference: mean of the two view losses
        ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        (0.5 * (ref(xs[0]).pow(2).mean() + ref(xs[1]).pow(2).mean())).backward()
        ropt.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])                      # replicas stay bit-identical
        assert sfd.replicas_identical(net)
        # zero-copy bucket: grads are views of one flat buffer, one in-place all-reduce, same update as above;
        # a parameter that got no gradient on one rank still takes part (zeros) -- equal message sizes by construction
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
        net2.load_state_dict({**{k: v for k, v in ref.state_dict().items()}, "2.weight": torch.eye(4), "2.bias": torch.zeros(4)})
        bucket = sfd.FlatGradBucket(net2.parameters())
        opt2 = torch.optim.Adam(net2.parameters(), lr=1e-2)
        for it in range(2):
            bucket.zero()
            y = net2[1](net2[0](xs[rank]))
            (net2[2](y) if rank == 0 else y).pow(2).mean().backward()      # rank 1 never touches layer 2
            assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net2.parameters())    # still views of the buffer
            bucket.all_reduce()
            opt2.step()
            assert sfd.replicas_identical(net2)
        # a re-bound .grad (torch's zero_grad(set_to_none=True) default) is refused instead of reducing a stale buffer
        opt2.zero_grad()
        for fn in (bucket.all_reduce, bucket.zero):
            try:
                fn()
                raise AssertionError("re-bound gradients accepted")
            except RuntimeError as e:
                assert "re-bound" in str(e)
        bucket.bind()
        bucket.zero()
        # the bucket on the REAL field's parameter list (CPU tensors: shapes and layout only) with the 8-GPU view split of
        # BASELINE configs[3] (32 novel views, 4 per GPU): one 7.46 MB collective, identical layout on every rank
        from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
        torch.manual_seed(0)
        field = NeRFNetwork(get_default_torch_ngp_opt())
        fb = sfd.FlatGradBucket(field.parameters())
        assert fb.flat.numel() == sum(p.numel() for p in field.parameters()) == 929336 * 2 + 6532
        assert [len(sfd.shard_views(32, r, 8)) for r in range(8)] == [4] * 8
        assert sum((sfd.shard_views(32, r, 8) for r in range(8)), []) == list(range(32))
        fb.zero()
        for k, p_ in enumerate(field.parameters()):
            p_.grad.add_(float(rank + 1) * (k + 1))                  # "backward": accumulates INTO the views
        fb.all_reduce()
        for k, p_ in enumerate(field.parameters()):
            assert torch.all(p_.grad == 1.5 * (k + 1))                # mean over the two ranks, in place, still the same views
        fb.check_bound()
        try:
            sfd.FlatGradBucket([torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2, dtype=torch.float64))])
            raise AssertionError("mixed dtypes accepted")
        except ValueError:
            pass
        # a diverged replica is detected
        if rank == 1:
            with torch.no_grad():
                net2[0].bias[0] += 1e-6
        assert not sfd.replicas_identical(net2)
        # unequal shard sizes are refused when asked to check
        try:
            sfd.all_gather_latents(torch.zeros(1 + rank, 4, 2, 2), check=True)
            raise AssertionError("ragged all-gather accepted")
        except RuntimeError:
            pass
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_view_sharding_gloo_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_shard_views_partition():
    from sparsefusion_amd.distributed import shard_views
    for n, w in ((32, 8), (7, 2), (3, 4), (0, 2)):
        parts = [shard_views(n, r, w) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_views(32, 3, 8) == [12, 13, 14, 15]               # BASELINE config 4: 4 views per GPU


_ONE_RANK_GLOO = r"""
import torch, torch.distributed as dist
from sparsefusion_amd import distributed as D
dist.init_process_group("gloo", rank=0, world_size=1)
assert not D._no_exchange()                      # SF_DIST_SINGLE_RANK_COLLECTIVES=1: a one-rank group still issues its collectives
lat = torch.randn(2, 4, 32, 32)
out = D.all_gather_latents(lat, check=True)
assert out.data_ptr() != lat.data_ptr() and torch.equal(out, lat)
net = torch.nn.Linear(8, 4)
b = D.FlatGradBucket(net.parameters())
b.zero()
net(torch.randn(3, 8)).sum().backward()
ref = b.flat.clone()
w = b.all_reduce(async_op=True)
assert w is not None
w.wait()
assert torch.equal(b.flat, ref)
D.all_reduce_grads(list(net.parameters()))
D.broadcast_params(net)
assert D.replicas_identical(net) is True
dist.destroy_process_group()
print("ONE_RANK_OK")
"""


def test_one_rank_group_issues_its_collectives_on_request():
    """The switch the single-rank RCCL tests of tests/test_gpu_bench_multirank.py rely on, on gloo: without it a one-rank group
    short-cuts every helper (nothing to exchange), with it the collectives run and return their inputs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 90), SF_DIST_SINGLE_RANK_COLLECTIVES="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", _ONE_RANK_GLOO], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ONE_RANK_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    env.pop("SF_DIST_SINGLE_RANK_COLLECTIVES")
    out = subprocess.run([sys.executable, "-c", _ONE_RANK_GLOO.replace("assert not D._no_exchange()", "assert D._no_exchange(); print('ONE_RANK_OK'); raise SystemExit(0)")],
                         env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ONE_RANK_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
