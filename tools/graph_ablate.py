"""Marginal cost of each op class of the B-sample UNet eval INSIDE the replayed hipGraph: the body is captured with one class of ops
left out (the values downstream are then garbage -- no kernel has data-dependent control flow, so the timing of the others stands)
and the difference to the full graph is that class's cost as the sampler pays it.  A rocprofv3 kernel trace of plain launches
overstates small kernels (dispatch + profiling overhead inside the duration); this does not.   usage: graph_ablate.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd import _lib
from sparsefusion_amd.unet import Unet, OP_CONV, OP_FCONV, OP_SLOTS, OP_GCA, OP_LN, OP_ATTN, OP_SPLITK_REDUCE, OP_INITX, OP_ELTWISE, FNORM_GN_SELF, FNORM_GN_SLOTS, FNORM_LN, FNORM_ATTN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
for kv in [a for a in os.environ.get("SF_UNET_ATTRS", "").split(",") if a]:
    setattr(unet, kv.split("=")[0], int(kv.split("=")[1]))
x, cond = torch.randn(B, 4, 32, 32, device=dev), torch.randn(B, 256, 32, 32, device=dev)
ctx = unet.begin_sampling(cond, torch.linspace(-3, 3, 4, device=dev))
unet.eval_prepared(ctx, x, 0)
plan = ctx["plan"]
ops = [plan.body_array[k] for k in range(plan.n_body_ops)]
lib = _lib.lib()


def cls(k):
    o = ops[k]
    if k and ops[k - 1].type == OP_FCONV and ops[k - 1].flags & 16:
        return cls(k - 1)                                   # the res_conv half of a pair belongs to its first half
    if o.type == OP_FCONV:
        H, norm = o.i[1], o.i[12]
        tag = {FNORM_GN_SELF: "gn_self", FNORM_GN_SLOTS: "gn_slots", FNORM_LN: "ln", FNORM_ATTN: "attn"}.get(norm, "plain")
        return f"fconv_{H}x{H}_{tag}" + ("_pipe" if o.flags & 32 else "") + ("_pool" if o.flags & 64 else "") + ("_pair" if o.flags & 16 else "")
    if o.type == OP_GCA:
        return "gca_" + {1: "pool", 2: "net0", 3: "gate"}.get(o.flags, str(o.flags)) + f"_{int(round((o.i[2] if o.flags != 2 else 0) ** 0.5))}"
    if o.type == OP_CONV:
        return f"igemm_{o.i[1]}x{o.i[2]}_k{o.i[9]}" + ("_pixshuf" if o.flags & 2 else "") + ("_deferred" if o.flags & 8 else "")
    return {OP_SLOTS: "slots", OP_LN: "layernorm", OP_ATTN: "attn16", OP_SPLITK_REDUCE: "splitk_reduce", OP_INITX: "init_x", OP_ELTWISE: "eltwise"}.get(o.type, f"op{o.type}")


def graph_ms(keep, reps=40):
    sub = (_lib.SfOp * len(keep))(*[ops[k] for k in keep])
    run = lambda: _lib.check(lib.sf_plan_run(sub, len(keep), _lib.stream_ptr()), "sub-plan")
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


allk = list(range(len(ops)))
full = graph_ms(allk)
names = {}
for k in allk:
    names.setdefault(cls(k), []).append(k)
print(f"B={B} body ops {len(ops)}  full graph {full:.1f} us")
tot = 0.0
rows = []
for name, ks in names.items():
    t = graph_ms([k for k in allk if k not in set(ks)])
    n_launch = sum(1 for k in ks if not (k and ops[k - 1].type == OP_FCONV and ops[k - 1].flags & 16))
    alone = graph_ms(ks) if len(ks) > 1 or True else 0.0
    rows.append((full - t, name, n_launch, alone))
    tot += full - t
for d, name, n, alone in sorted(rows, reverse=True):
    print(f"  {name:36s} launches {n:3d}  marginal {d:7.1f} us ({d / n:5.2f} each)   alone {alone:7.1f} us ({alone / n:5.2f} each)")
print(f"  sum of marginals {tot:.1f} us")
