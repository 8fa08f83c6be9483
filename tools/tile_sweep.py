"""Measured tile picks for the implicit-GEMM launches the cost model of Unet.conv_tiling gets wrong: for every k_conv_igemm
layer of the canonical B = 1 eval body, try every valid (WM, WN, split-K groups) and time the whole eval (graph replay, as the
sampler runs it); prints the winners as a dict literal for Unet.TILE_PICKS.    python tools/tile_sweep.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd.unet import Unet, OP_CONV
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
x, cond = torch.randn(B, 4, 32, 32, device=dev), torch.randn(B, 256, 32, 32, device=dev)
unet.tile_override = {}


def eval_ms(n=150):
    unet.drop_plans()
    ctx = unet.begin_sampling(cond, torch.linspace(-3, 3, 8, device=dev))
    for k in range(8):
        unet.eval_prepared(ctx, x, k % 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        unet.eval_prepared(ctx, x, k % 8)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, ctx["plan"]


base, plan = eval_ms()
print(f"B={B} baseline eval {base:.4f} ms", flush=True)
layers = []
for k in range(plan.n_body_ops):
    o = plan.body_array[k]
    if o.type == OP_CONV and o.i[14] < 256:
        Bq, H, W, Cin, Ho, Wo, Cout, kk = o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.i[9]
        key = ((Bq * Ho * Wo + 15) // 16, (Cout + 15) // 16, kk * kk * (Cin // 32), bool(o.flags & 2))
        if key not in [l[0] for l in layers]:
            layers.append((key, (o.i[14] // 16, o.i[14] % 16, o.i[13])))
picks = {}
for key, cur in layers:
    m_frags, n_frags, KS, pix = key
    res = []
    for WM in (1, 2, 4):
        if m_frags % WM:
            continue
        for WN in (1, 2, 4):
            if n_frags % WN:
                continue
            for groups in ((1,) if pix else (1, 2, 4, 8, 16)):
                if groups > 1 and KS // (4 * groups) < 2:
                    continue
                unet.tile_override = dict(picks)
                unet.tile_override[key] = (WM, WN, groups)
                try:
                    ms, _ = eval_ms(100)
                except Exception as e:                       # a combination the kernel rejects
                    ms = float("inf")
                res.append((ms, (WM, WN, groups)))
    res.sort()
    best_ms, best = res[0]
    cur_ms = [m for m, c in res if c == cur]
    print(f"layer m_frags={m_frags} n_frags={n_frags} KS={KS} pixshuf={pix}: model pick {cur} {cur_ms[0] if cur_ms else float('nan'):.4f} ms -> best {best} {best_ms:.4f} ms   "
          f"[{', '.join(f'{c}:{m:.4f}' for m, c in res[:4])}]", flush=True)
    if cur_ms and best_ms < cur_ms[0] - 0.002:              # keep a pick only when it wins by more than the timing noise
        picks[key] = best
unet.tile_override = dict(picks)
final, _ = eval_ms()
print(f"B={B} with picks {final:.4f} ms (baseline {base:.4f})")
print("TILE_PICKS =", picks)
