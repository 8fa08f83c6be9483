"""Side-by-side of tools/vae_layers.py logs: python tools/vae_cmp.py a.log b.log ..."""
import re, sys
def load(f):
    d, kind = {}, None
    for l in open(f):
        if l.startswith('=='):
            kind = l.split()[1].rstrip(':'); continue
        m = re.match(r'\s*(\d+x\d+\s+\d+->\s*\d+ k\d out \S+ tile\s+\d+ a_f32=\d)\s+x\s*(\d+)\s+([\d.]+) ms\s+([\d.]+)', l)
        if m: d[(kind, m.group(1))] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return d
logs = [load(f) for f in sys.argv[1:]]
a = logs[0]
for k in sorted(a, key=lambda k: -a[k][1]):
    if a[k][1] < 0.03: continue
    print(f"{k[0]} {k[1]:56s} x{a[k][0]:2d} " + " | ".join(f"{l[k][1]:.3f} ms {l[k][2]:6.1f} TF" for l in logs if k in l))
