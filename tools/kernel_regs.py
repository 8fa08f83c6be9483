"""Register / LDS / scratch use of every kernel of one HIP source, from the gfx950 assembly hipcc emits (no GPU needed).
usage: kernel_regs.py [source.hip] [name filter] [-D...] [--mfma-overlaps]      e.g.  kernel_regs.py unet_fused.hip pipe
--mfma-overlaps: per kernel, the MFMAs whose destination PARTIALLY overlaps a source (kind = operand, direction of the destination's start
relative to the source's, distance): "C+2" is the kind that returned wrong sums on gfx950 (csrc/sf_dev.h sf_mfma16_acc)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsefusion_amd import build as B
OVER = "--mfma-overlaps" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("-")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
src = args[0] if args else "unet_fused.hip"
flt = args[1] if len(args) > 1 else ""
import zlib
out = "/tmp/%s.%d.s" % (os.path.basename(src), zlib.crc32(" ".join(defs).encode()) % 9999)
if not os.path.exists(out) or os.path.getmtime(out) < B._deps_mtime(): subprocess.check_call([B.HIPCC] + B.FLAGS + defs + ["--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out])
txt = open(out).read()
dem = {}
names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
if names:
    d = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    dem = dict(zip(names, d))
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\." + k + r" (\S+)", body) or [None, "?"])[1]
    dn = dem.get(name, name)
    if flt and flt not in dn:
        continue
    meta = re.search(r"\.name:\s+" + re.escape(name) + r"\b(.*?)(?=\n  - \.|\Z)", txt, re.S)
    ms = meta.group(1) if meta else ""
    mg = lambda k: (re.search(r"\." + k + r":\s+(\S+)", ms) or [None, "?"])[1]
    print(f"vgpr {g('amdhsa_next_free_vgpr'):>4s} accum_off {g('amdhsa_accum_offset'):>4s} sgpr {g('amdhsa_next_free_sgpr'):>4s} "
          f"scratch {mg('private_segment_fixed_size'):>5s} vspill {mg('vgpr_spill_count'):>3s} sspill {mg('sgpr_spill_count'):>3s}  {dn[:150]}")

if OVER:
    rng = lambda t: (lambda m: (int(m.group(1)), int(m.group(2))) if m else None)(re.match(r"[va]\[(\d+):(\d+)\]", t))
    cur, found = None, {}
    for ln in txt.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+v_mfma_\w+ (\S+), (\S+), (\S+), (\S+)", ln)
        if m and cur:
            d, *srcs = [rng(x.rstrip(",")) for x in m.groups()]
            for key, sr in zip("ABC", srcs):
                if sr and d and not (sr[1] < d[0] or d[1] < sr[0]) and sr != d:
                    found.setdefault(cur, []).append(f"{key}{'+' if d[0] > sr[0] else '-'}{abs(d[0] - sr[0])}")
    bad = 0
    for n, kinds in found.items():
        dn = dem.get(n, n)
        if flt and flt not in dn:
            continue
        bad += sum(k.startswith("C+") for k in kinds)
        print("mfma partial overlaps", kinds, dn[:150])
    print(f"{bad} MFMAs with a destination that starts above an overlapping SrcC")
    sys.exit(1 if bad else 0)
