"""View sharding across the GPUs of one node.

The reference runs whole scenes per rank and issues no collective on the distillation path
(demo.py:22,59).  The MI355X design shards the independent NOVEL VIEWS of one distillation step:
rank r owns views [r*V, (r+1)*V); every rank keeps a replica of the NGP field, so two exchanges keep
the replicas identical and let every rank see the whole batch of rendered latents:

  * all_gather_latents   [V,4,32,32] per rank -> [world*V,4,32,32]     (RCCL all-gather, 64 KiB fp32 per rank)
  * all_reduce_grads     mean of the NGP gradients (7.43 MB table + 26 KB MLP) as ONE flat all-reduce
                         before each optimizer.step

torch.distributed's "nccl" backend IS RCCL on ROCm (xGMI inside a node); the same code runs on "gloo"
for the CPU tests.  Messages are small (<= 8 MB), so one flat buffer per step is the right granularity
for the point-to-point xGMI fabric (per-link bound ring of 7 hops would dominate otherwise)."""
import os

import torch
import torch.distributed as dist


def _no_exchange(group=None):
    """True when there is nobody to exchange with: no process group, or a group of one rank.  SF_DIST_SINGLE_RANK_COLLECTIVES=1
    makes a one-rank group issue its collectives anyway -- a 1-GPU box can then put RCCL under every call site of this module
    (`tests/test_gpu_bench_multirank.py::test_single_rank_rccl`); results are the inputs, by definition of the collectives."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("SF_DIST_SINGLE_RANK_COLLECTIVES") != "1"


def shard_views(n_views, rank, world):
    """Contiguous block partition of view indices; the first (n_views % world) ranks get one extra."""
    base, extra = divmod(n_views, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def all_gather_latents(latents, group=None, check=False):
    """[V, C, H, W] on every rank -> [world*V, C, H, W] in rank order.  V must be the same on every rank
    (`all_gather_into_tensor` has no ragged form): pad the short shards of an uneven `shard_views` split, or pass
    check=True to have the ranks compare their V first (one tiny extra collective)."""
    if _no_exchange(group):
        return latents
    world = dist.get_world_size(group)
    if check:
        v = torch.tensor([latents.shape[0], -latents.shape[0]], device=latents.device, dtype=torch.int64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
        if int(v[0]) != -int(v[1]):
            raise RuntimeError(f"all_gather_latents: ranks hold between {-int(v[1])} and {int(v[0])} views; pad to equal counts")
    out = latents.new_empty((world * latents.shape[0],) + tuple(latents.shape[1:]))
    dist.all_gather_into_tensor(out, latents.contiguous(), group=group)
    return out


class FlatGradBucket:
    """The gradients of `params` as views of ONE persistent flat buffer: the all-reduce of a step is a single in-place
    collective on that buffer -- no gather, no scatter, no allocation (the first round concatenated 7.46 MB, reduced,
    divided and copied back, twice per step).  Every rank reduces the same layout whether or not a parameter received a
    gradient this step (an unused parameter contributes zeros), so replicas can never disagree on the message size.

        bucket = FlatGradBucket(ngp.parameters())
        bucket.zero()            # instead of optimizer.zero_grad(): autograd then accumulates INTO the views
        loss.backward()
        bucket.all_reduce()      # mean over the replicas, in place
        optimizer.step()
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no parameter requires a gradient")
        ref = self.params[0]
        for p in self.params:                                       # one collective on one buffer: one dtype, one device
            if p.dtype != ref.dtype or p.device != ref.device:
                raise ValueError(f"FlatGradBucket: parameters must share dtype and device (got {p.dtype} on {p.device} and "
                                 f"{ref.dtype} on {ref.device})")
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.bind()

    def bind(self):
        """(Re-)attach every .grad to its slice of the flat buffer.  Needed again after anything that REBINDS .grad:
        optimizer.zero_grad() (set_to_none=True is the torch default), `p.grad = None`, module.to() / _apply."""
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def check_bound(self):
        """Raise if some .grad is no longer the bucket's view: the collective would then reduce a stale buffer while the
        optimizer steps on un-reduced per-rank gradients -- replicas would diverge without any error."""
        isz = self.flat.element_size()
        base = self.flat.data_ptr()
        for k, (p, off) in enumerate(zip(self.params, self.offsets)):
            g = p.grad
            if g is None or g.data_ptr() != base + off * isz or g.shape != p.shape or not g.is_contiguous():
                raise RuntimeError(f"FlatGradBucket: .grad of parameter {k} (shape {tuple(p.shape)}) was re-bound away from the flat "
                                   "buffer (optimizer.zero_grad(set_to_none=True)? module.to()?): use bucket.zero() instead of "
                                   "zero_grad(), or call bucket.bind() after moving the module")

    def zero(self):
        self.check_bound()
        self.flat.zero_()

    def all_reduce(self, group=None, average=True, async_op=False):
        """In-place sum (mean) over the group; returns the work handle when async_op (wait before optimizer.step)."""
        self.check_bound()
        if _no_exchange(group):
            return None
        if average:
            self.flat.div_(dist.get_world_size(group))           # before the sum: same result, and the async form needs no epilogue
        return dist.all_reduce(self.flat, group=group, async_op=async_op)


def replicas_identical(module, group=None):
    """True when every rank holds bit-identical parameters: MAX and MIN over the ranks of an fp64 checksum and of the
    parameter extrema agree.  Two 4-element collectives; meant for a per-step assertion in multi-GPU runs."""
    if _no_exchange(group):
        return True
    ps = [p.detach() for p in module.parameters()]
    sig = torch.stack([sum(p.double().sum() for p in ps), sum((p.double() ** 2).sum() for p in ps),
                       max(p.max() for p in ps).double(), min(p.min() for p in ps).double()])
    hi, lo = sig.clone(), sig.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return bool(torch.equal(hi, lo))


def all_reduce_grads(params, group=None, average=True):
    """In-place mean (or sum) of the .grad of `params` over the group with ONE flat collective (generic form: gathers into
    a temporary; the hot path uses FlatGradBucket).  Parameters without a gradient contribute zeros, so every rank
    reduces the same layout."""
    if _no_exchange(group):
        return
    params = [p for p in params if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def broadcast_params(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters (one flat broadcast)."""
    if _no_exchange(group):
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in ps:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
