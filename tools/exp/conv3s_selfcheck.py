"""Debug build (C3S_DBG=1024) of k_conv3s: every staging thread re-derives the element it just stored (other instruction sequence) and reads the
LDS location back; mismatches are counted / recorded in the op's debug buffer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fused_cases as fc
name = sys.argv[1]
from hostemu import fused
cap = {}
orig_mkop = fused.mkop
def mkop(type_, flags=0, p=(), i=(), f=()):
    if len(p) > 17 and p[17] is not None and (flags & 64):
        cap["weff"] = p[17]
    return orig_mkop(type_, flags, p, i, f)
fused.mkop = mkop
fc.fused.mkop = mkop
for rep in range(2):
    dbg = torch.zeros(8 + 30 * 12 + 64, dtype=torch.float32, device="cuda:0")
    try:
        e = fc.run_conv_case("gpu", **dict(fc.CONV_CASES_FULL[name], tol=1e9, dbg=dbg))
    except AssertionError as ex:
        e = str(ex)
    n = int(dbg[:2].view(torch.int64)[0])
    print(f"rep {rep}: rel {e}; mismatching staged elements: {n}; double-read differences {int(dbg[400:402].view(torch.int64)[0])}")
    rec = dbg[2:2 + 360].view(-1, 12).cpu()
    we = cap["weff"].float().cpu() if "weff" in cap else None
    for k in range(min(n, 30)):
        if we is not None:
            hit = (we - float(rec[k][4])).abs() < 1e-6
            print("      value found in the w_eff table at element(s)", hit.nonzero().flatten().tolist()[:6], "of", we.numel())
        r = rec[k]
        print(f"   tid {int(r[0])} (wave {int(r[0]) >> 6} lane {int(r[0]) & 63}) wg {int(r[10])} e {int(r[1])} chunk {int(r[2])} comp {int(r[3])}: [4] {r[4]:.6f} [5] {r[5]:.6f} [6] {r[6]:.6f} [7] {r[7]:.6f} | two chunks ago {r[6]:.5f} other buffer now {r[9]:.5f} | loff {int(r[11])}")
