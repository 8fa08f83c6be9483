"""HBM fetch of the UNet eval's fused-conv launches by kernel class and by source (r05, the r04 review's item 5).

Input: rocprofv3 --kernel-trace --pmc FETCH_SIZE counter CSVs of `SF_FCX_PLAIN=<variant> tools/fconv4_knockout.py 1` runs (N plain evals; that driver and its phase knock-out builds of the r05 headers were retired in r06 -- `git show c0adaa6:tools/fconv4_knockout.py`):
  argv[1] = directory of the PRODUCT run, argv[2] = directory of the run whose 15 4x4 GroupNorm-self launches load no weights (variant 1),
  argv[3] (optional) = calibration factor of FETCH_SIZE on nt dwordx4 weight streams (tools/exp/weight_prefetch_chain.hip under the counter).
The dispatches of an eval are joined with the plan's launches BY POSITION (between two k_init_x dispatches, copies skipped), every launch is
labelled with tools/graph_ablate.py's class and carries the weight bytes its op(s) stream (algorithmic)."""
import csv
import glob
import os
import sys
from collections import defaultdict

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsefusion_amd.unet import Unet, OP_CONV, OP_FCONV, OP_GCA, FNORM_GN_SELF, FNORM_GN_SLOTS, FNORM_LN, FNORM_ATTN


def plan_launches():
    """(class, algorithmic weight bytes) of every kernel dispatch of one B = 1 eval body, in order.  The plan is built in SIZING mode on the
    CPU (fake pointers, same op list): the tool runs without a GPU on counter CSVs copied back from the box."""
    from sparsefusion_amd.unet import _Plan
    unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    plan = _Plan(unet, 1, torch.device("cpu")).build()
    ops = [plan.body_array[k] for k in range(plan.n_body_ops)]

    def wbytes(o):
        if o.type == OP_FCONV:
            return 2 * o.i[5] * (o.i[3] + o.i[4]) * o.i[8] * o.i[8]
        if o.type == OP_CONV:
            return 2 * o.i[6] * o.i[3] * o.i[9] * o.i[10]
        return 0

    def cls(o):
        if o.type == OP_FCONV:
            H, norm = o.i[1], o.i[12]
            tag = {FNORM_GN_SELF: "gn_self", FNORM_GN_SLOTS: "gn_slots", FNORM_LN: "ln", FNORM_ATTN: "attn"}.get(norm, "plain")
            return f"fconv_{H}x{H}_{tag}" + ("_pipe" if o.flags & 32 else "") + ("_pair" if o.flags & 16 else "")
        if o.type == OP_CONV:
            return f"igemm_{o.i[1]}x{o.i[2]}"
        if o.type == OP_GCA:
            return "gca"
        return "other"
    out, k = [], 0
    while k < len(ops):
        o = ops[k]
        if o.type == 8:                                         # OP_MEMSET: hipMemsetAsync, not one of the library's kernels
            k += 1
            continue
        if o.type == OP_FCONV and o.flags & 16:
            out.append((cls(o), wbytes(o) + wbytes(ops[k + 1])))
            k += 2
        else:
            out.append((cls(o), wbytes(o)))
            if o.type == OP_CONV and o.i[13] > 1 and not (o.flags & 8):
                out.append(("splitk_reduce_of_" + cls(o), 0))       # a split-K implicit GEMM that is not deferred reduces in its own launch
            k += 1
    return out


def evals_of(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))     # KiB -> bytes
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_init_x" in r[1]]
    evs = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        evs.append([r for r in rows[a:b] if r[1].lstrip().startswith(("k_", "void k_"))])      # the library's kernels only (no copies / fills / torch kernels)
    return evs


launches = plan_launches()
cal = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
tab = {}
for tag, d in (("product", sys.argv[1]), ("noweights_4x4", sys.argv[2])):
    evs = [e for e in evals_of(d) if len(e) == len(launches)]
    if not evs:
        allev = evals_of(d)
        lens = sorted({len(e) for e in allev})
        print(f"{tag}: no eval with {len(launches)} dispatches (seen {lens})")
        if allev:
            e = max(allev, key=len)
            for i in range(max(len(e), len(launches))):
                print(f"   {i:3d} {launches[i][0] if i < len(launches) else '-':32s} {e[i][1][:60] if i < len(e) else '-'}")
        continue
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    for e in evs[1:] or evs:                                    # (the first eval of a process is cold in every cache)
        for (c, wb), (_, name, fetch) in zip(launches, e):
            acc[c][0] += fetch
            acc[c][1] += wb
            acc[c][2] += 1
    n = len(evs[1:] or evs)
    tab[tag] = {c: (v[0] / n, v[1] / n, v[2] // n) for c, v in acc.items()}
p = tab.get("product", {})
q = tab.get("noweights_4x4", {})
print(f"per eval (B = 1), FETCH_SIZE in MB as counted (x1) and x{cal:.2f} (calibration on nt dwordx4 weight streams); algorithmic = bf16 weight bytes of the class's ops")
print(f"{'class':34s} {'launches':>8s} {'algorithmic':>12s} {'fetch x1':>10s} {'fetch cal':>10s} {'cal / alg':>9s}")
tot = [0.0, 0.0]
for c, (f, wb, nl) in sorted(p.items(), key=lambda kv: -kv[1][0]):
    print(f"{c:34s} {nl:8d} {wb / 1e6:12.1f} {f / 1e6:10.1f} {f * cal / 1e6:10.1f} {(f * cal / wb if wb else float('nan')):9.2f}")
    if c.startswith("fconv"):
        tot[0] += f
        tot[1] += wb
if tot[1]:
    print(f"{'all fused-conv launches':34s} {'':8s} {tot[1] / 1e6:12.1f} {tot[0] / 1e6:10.1f} {tot[0] * cal / 1e6:10.1f} {tot[0] * cal / tot[1]:9.2f}")
if "fconv_4x4_gn_self" in p and "fconv_4x4_gn_self" in q:
    f, wb, nl = p["fconv_4x4_gn_self"]
    f0 = q["fconv_4x4_gn_self"][0]
    print(f"\n4x4 GroupNorm-self launches by source ({nl} launches, per eval): activations + parameters {f0 / 1e6:.1f} MB as counted (these launches without their weight "
          f"loads), weights {(f - f0) / 1e6:.1f} MB as counted = {(f - f0) * cal / 1e6:.1f} MB calibrated against {wb / 1e6:.1f} MB algorithmic "
          f"({(f - f0) * cal / wb:.2f}x); activation share of the class's fetch {f0 / f:.1%} (x1 scale)")
