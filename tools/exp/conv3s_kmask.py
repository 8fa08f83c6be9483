"""Which k-steps of a failing k_conv3s launch are wrong?  Runs one fused_cases conv case on k_conv3s and on the general pipelined kernel with
the weights masked to ONE 128-channel chunk (and one tap) at a time and prints the relative difference of the two outputs per (chunk, tap)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fused_cases as fc
from hostemu import fused
name = sys.argv[1]
kw = dict(fc.CONV_CASES_FULL[name], tol=1e9)
C = kw["C1"]
orig_pack = fused.pack_conv_weights
orig_rel = fc.rel
def run(mask, keep_pipe):
    def pack(w):
        if w.shape[-1] == 3:
            w = w * mask
        return orig_pack(w)
    fused.pack_conv_weights = pack
    cap = []
    def rel2(a, b):
        cap.append(a.clone())
        return 0.0
    fc.rel = rel2
    try:
        fc.run_conv_case("gpu", **dict(kw, keep_pipe=keep_pipe))
    except AssertionError:
        pass
    fc.rel = orig_rel
    fused.pack_conv_weights = orig_pack
    return cap[0]
for c in range(C // 128):
    row = []
    for tap in (None, 0, 4, 8):
        m = torch.zeros(1, C, 3, 3)
        if tap is None:
            m[:, c * 128:(c + 1) * 128] = 1
        else:
            m[:, c * 128:(c + 1) * 128, tap // 3, tap % 3] = 1
        a, b = run(m, False), run(m, True)
        row.append(f"{'all' if tap is None else tap}: {orig_rel(a, b):.1e}")
    print(f"chunk {c}:", "  ".join(row), flush=True)
for sub in range(4):
    m = torch.zeros(1, C, 3, 3)
    for c in range(C // 128):
        m[:, c * 128 + sub * 32:c * 128 + sub * 32 + 32] = 1
    print(f"sub-chunk {sub} of every chunk: {orig_rel(run(m, False), run(m, True)):.1e}", flush=True)
