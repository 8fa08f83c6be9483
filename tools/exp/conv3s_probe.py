"""Read the staged A operand of a k_conv3s launch back through the conv itself: weights = a selector that copies input channel c0 + n of the
centre tap to output channel n, so out[p][n] = SiLU(GroupNorm(x))[p][c0 + n] as the kernel staged it.  Compared with the general kernel."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fused_cases as fc
from hostemu import fused
name = sys.argv[1]
kw = dict(fc.CONV_CASES_FULL[name], tol=1e9)
C = kw["C1"]
orig_pack, orig_rel = fused.pack_conv_weights, fc.rel
def run(c0, keep_pipe, tap=4):
    def pack(w):
        if w.shape[-1] == 3:
            w = torch.zeros_like(w)
            for n in range(32):
                w[n, c0 + n, tap // 3, tap % 3] = 1.0
        return orig_pack(w)
    fused.pack_conv_weights = pack
    cap = []
    fc.rel = lambda a, b: (cap.append(a.clone()), 0.0)[1]
    try:
        fc.run_conv_case("gpu", **dict(kw, keep_pipe=keep_pipe))
    except AssertionError:
        pass
    fc.rel, fused.pack_conv_weights = orig_rel, orig_pack
    return cap[0][:, :32]
for c0 in [c for c in (0, 32, 64, 96, 128, 160, 192, 224, 256, 256 + 32, 256 + 64, 256 + 96, 384 + 64) if c + 32 <= C]:
    a, b = run(c0, False), run(c0, True)
    d = (a - b).abs()
    bad = (d > 1e-6 * b.abs().max()).float()
    print(f"channels {c0}..{c0 + 31}: rel {orig_rel(a, b):.2e}; wrong elements {int(bad.sum())} of {bad.numel()}; per channel {bad.sum(0).int().tolist()}")
    if bad.sum() > 0:
        idx = bad.nonzero()[:6]
        for p, n in idx.tolist():
            print(f"    pixel {p} channel {c0 + n}: conv3s {a[p, n]:.6f} general {b[p, n]:.6f} ratio {a[p, n] / b[p, n]:.5f}")
        print("    wrong pixels:", sorted(set(bad.nonzero()[:, 0].tolist()))[:64])
