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
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world):
    """Contiguous block partition of view indices; the first (n_views % world) ranks get one extra."""
    base, extra = divmod(n_views, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def all_gather_latents(latents, group=None):
    """[V, C, H, W] on every rank -> [world*V, C, H, W] in rank order (equal V on all ranks)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return latents
    world = dist.get_world_size(group)
    out = latents.new_empty((world * latents.shape[0],) + tuple(latents.shape[1:]))
    dist.all_gather_into_tensor(out, latents.contiguous(), group=group)
    return out


def all_reduce_grads(params, group=None, average=True):
    """In-place mean (or sum) of the .grad of `params` over the group with ONE flat collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
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
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in ps:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
