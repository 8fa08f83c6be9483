"""Register / LDS / scratch use of every kernel of one HIP source, from the gfx950 assembly hipcc emits (no GPU needed).
usage: kernel_regs.py [source.hip] [name filter] [-D...]      e.g.  kernel_regs.py unet_fused.hip pipe"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsefusion_amd import build as B
args = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
src = args[0] if args else "unet_fused.hip"
flt = args[1] if len(args) > 1 else ""
out = f"/tmp/{os.path.basename(src)}.{abs(hash(tuple(defs))) % 9999}.s"
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
