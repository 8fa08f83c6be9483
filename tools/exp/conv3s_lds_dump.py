"""Debug build C3S_DBG=8192: every workgroup of a k_conv3s launch dumps its two frame buffers after the chunk loop; here they are compared with
the frames the op should have staged (torch: GroupNorm -> scale/shift -> SiLU -> bf16) -- which locations are wrong, and does a wrong value equal
the right value of some OTHER location (a misdirected store) or of another chunk (a lost store)?   8x8 maps, full-width strips only."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fused_cases as fc
from hostemu import fused
name = sys.argv[1]
kw = dict(fc.CONV_CASES_FULL[name], tol=1e9)
B, H, W, C = kw["B"], kw["H"], kw["W"], kw["C1"]
assert B == 1 and H == 8 and W == 8
TR, FR, FW, PSTR = 2, 4, 10, 288
BUF = ((FR * FW + 1) * PSTR + 15) // 16 * 16
NCH = C // 128
MT, NTILES = H // TR, C // 16
grid = MT * NTILES
cap = {}
orig_mkop = fused.mkop
def mkop(type_, flags=0, p=(), i=(), f=()):
    if type_ == fc.OP_FCONV and (flags & 32):
        cap.update(x=p[0], gamma=p[13], beta=p[14], ss=p[15])
    return orig_mkop(type_, flags, p, i, f)
fused.mkop = mkop
dump = torch.zeros(grid * 2 * BUF // 4, dtype=torch.int32, device="cuda:0")
try:
    e = fc.run_conv_case("gpu", **dict(kw, dbg=dump))
except AssertionError as ex:
    e = str(ex)
print("rel", e)
x = cap["x"].float().cpu().view(1, H, W, C).permute(0, 3, 1, 2)
y = F.group_norm(x, 8, cap["gamma"].cpu(), cap["beta"].cpu(), eps=1e-5)
if cap["ss"] is not None:
    ss = cap["ss"].cpu()
    y = y * (ss[:, :C, None, None] + 1) + ss[:, C:, None, None]
act = F.silu(y).to(torch.bfloat16).float()[0]                 # [C][H][W]
d = dump.cpu().view(grid, 2, BUF // 4).view(torch.bfloat16).float().view(grid, 2, BUF // 2)
tot = bad = 0
examples = []
for wg in range(grid):
    x8, j = wg & 7, wg >> 3
    mt = j % MT
    row0 = mt * TR
    for buf in range(2):
        c = NCH - 2 + buf if (NCH - 2) % 2 == 0 else NCH - 1 - buf      # last two chunks: chunk c sits in buffer c & 1
        c = [cc for cc in (NCH - 2, NCH - 1) if cc % 2 == buf][0]
        fr = d[wg, buf, :FR * FW * PSTR // 2].view(FR * FW, PSTR // 2)[:, :128]          # [frame pixel][128 channels]
        for fpx in range(FR * FW):
            r, xx = row0 - 1 + fpx // FW, fpx % FW - 1
            want = act[c * 128:(c + 1) * 128, r, xx] if (0 <= r < H and 0 <= xx < W) else torch.zeros(128)
            got = fr[fpx]
            m = (got - want).abs() > 1e-6
            tot += 128
            nb = int(m.sum())
            bad += nb
            if nb and len(examples) < 12:
                ch = m.nonzero().flatten().tolist()
                # where else does the wrong value occur (same channel, any pixel / the chunk two earlier)?
                c0 = ch[0]
                v = float(got[c0])
                same_chunk = ((act[c * 128 + c0] - v).abs() < 1e-6).nonzero().tolist()
                prev_chunk = ((act[(c - 2) * 128 + c0] - v).abs() < 1e-6).nonzero().tolist() if c >= 2 else []
                examples.append((wg, mt, buf, c, fpx, r, xx, ch[:8], len(ch), v, float(want[c0]), same_chunk[:3], prev_chunk[:3]))
print(f"{bad} wrong of {tot} frame elements")
for ex in examples:
    print("wg %d (tile %d) buffer %d chunk %d frame pixel %d (row %d col %d): wrong local channels %s.. (%d); first: got %.5f want %.5f; same value at pixels %s of this chunk / %s of chunk - 2" % ex)
