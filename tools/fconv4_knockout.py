"""Phase knock-outs of the 4x4 GroupNorm-self conv launches inside the replayed eval graph (tools/exp/fconv4_knockout.hip, r05).
For every variant: the B-sample eval body is captured with the 15 `fconv_4x4_gn_self` launches replaced by the variant's kernel (the
values downstream are garbage; no kernel has data-dependent control flow) -> eval time; and a graph of those launches alone.
usage: fconv4_knockout.py [B] [variant ...]"""
import ctypes as C
import os
import subprocess
import sys

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsefusion_amd import _lib
from sparsefusion_amd.unet import Unet, OP_FCONV, FNORM_GN_SELF

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = [v for v in sys.argv[2:]] or ["0", "1", "2", "3", "4", "5", "6", "7", "8", "9"]       # "<n>" or "<n>w<waves>"
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
x, cond = torch.randn(B, 4, 32, 32, device=dev), torch.randn(B, 256, 32, 32, device=dev)
ctx = unet.begin_sampling(cond, torch.linspace(-3, 3, 4, device=dev))
unet.eval_prepared(ctx, x, 0)
plan = ctx["plan"]
ops = [plan.body_array[k] for k in range(plan.n_body_ops)]
lib = _lib.lib()
sel = [k for k, o in enumerate(ops) if o.type == OP_FCONV and o.i[1] == 4 and o.i[12] == FNORM_GN_SELF and not (o.flags & (16 | 32))
       and not (k and ops[k - 1].type == OP_FCONV and ops[k - 1].flags & 16) and o.i[15] == 1 and o.i[16] == 1 and o.i[9] <= 1]
print(f"B={B}: {len(ops)} body ops, {len(sel)} 4x4 GroupNorm-self launches (lazy {sum(1 for k in sel if ops[k].i[9] == 1)})", flush=True)


def build(v, waves, defs=()):
    out = f"/tmp/fcx_{v}_{waves}_{abs(hash(tuple(defs))) % 100000}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", f"-DSF_FCX={v}",
                           f"-DSF_FCONV_WAVES={waves}", *["-D" + d for d in defs],
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "exp", "fconv4_knockout.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    l = C.CDLL(out)
    l.fcx_run.argtypes = [C.c_void_p, C.c_void_p]
    l.fcx_run.restype = C.c_int
    return l


def time_graph(run, reps=40):
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


one = [(_lib.SfOp * 2)(ops[k], ops[min(k + 1, len(ops) - 1)]) for k in range(len(ops))]
selset = set(sel)


def run_eval(vlib):
    st = _lib.stream_ptr()
    k = 0
    while k < len(ops):
        if k in selset and vlib is not None:
            rc = vlib.fcx_run(C.addressof(one[k]), st)
            assert rc == 0, (k, rc)
            k += 1
        elif ops[k].type == OP_FCONV and ops[k].flags & 16:
            _lib.check(lib.sf_plan_run(one[k], 2, st), "pair")
            k += 2
        else:
            _lib.check(lib.sf_plan_run(one[k], 1, st), "op")
            k += 1


def run_alone(vlib):
    st = _lib.stream_ptr()
    for k in sel:
        if vlib is None:
            _lib.check(lib.sf_plan_run(one[k], 1, st), "op")
        else:
            assert vlib.fcx_run(C.addressof(one[k]), st) == 0


if os.environ.get("SF_FCX_PLAIN"):        # SF_FCX_PLAIN=<variant>: N plain (un-captured) evals with that variant, for a rocprofv3 counter pass
    vs, *defs = os.environ["SF_FCX_PLAIN"].split(":")
    vl = build(int(vs), 8, defs) if vs != "product" else None
    for _ in range(int(os.environ.get("SF_FCX_EVALS", "3"))):
        run_eval(vl)
    torch.cuda.synchronize()
    sys.exit(0)
base_eval = time_graph(lambda: run_eval(None))
base_alone = time_graph(lambda: run_alone(None))
print(f"product library: eval {base_eval:8.1f} us   the {len(sel)} launches alone {base_alone:7.1f} us ({base_alone / len(sel):5.2f} each)", flush=True)
names = {0: "product kernel (this TU)", 1: "no weight loads", 2: "one load per lazy element (not six)", 3: "no statistics", 4: "no gamma/beta/scale-shift loads",
         5: "no SiLU", 6: "no MFMA", 7: "no cross-wave sum", 8: "empty body", 9: "no lazy materialisation"}
for vs in variants:
    vs, *defs = vs.split(":")                                   # "<n>[w<waves>][:MACRO=value ...]"
    v, waves = (int(vs.split("w")[0]), int(vs.split("w")[1])) if "w" in vs else (int(vs), 8)
    vl = build(v, waves, defs)
    te, ta = time_graph(lambda: run_eval(vl)), time_graph(lambda: run_alone(vl))
    nm = names.get(v, "mask " + ",".join(str(b) for b in range(16) if (v - 1000) >> b & 1) if v >= 1000 else str(v))
    nm += ("" if waves == 8 else f" [{waves} waves]") + (" " + " ".join(defs) if defs else "")
    print(f"variant {v:5d} {nm:64s} eval {te:8.1f} us ({(te - base_eval) / len(sel):+6.2f} per launch)   alone {ta:7.1f} us ({ta / len(sel):5.2f} each)", flush=True)
