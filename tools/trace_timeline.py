"""Timeline of the LAST UNet eval in a rocprofv3 kernel trace of tools/unet_eval_loop.py: per launch its duration and the
idle gap before it, plus per-kernel totals.  usage: trace_timeline.py <rocprof out dir> [out.txt]"""
import csv, glob, sys
from collections import defaultdict
trace = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_init_x' in r['Kernel_Name']]          # one per eval of the sampler path
if len(idx) < 2:
    idx = [i for i, r in enumerate(rows) if 'k_pack_in' in r['Kernel_Name']]     # the stand-alone forward
seg = rows[idx[-2]:idx[-1]] if len(idx) > 1 else rows[idx[-1]:]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
agg = defaultdict(lambda: [0, 0.0])
busy = 0.0
for k, r in enumerate(seg):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    d = (e - s) / 1e3
    busy += d
    agg[name][0] += 1
    agg[name][1] += d
    print(f"{k:4d} t={(s - t0) / 1e3:8.1f} gap={(s - prev_end) / 1e3:6.2f} dur={d:7.2f} grid={r.get('Grid_Size', '?'):>8s} wg={r.get('Workgroup_Size', '?'):>4s} "
          f"lds={r.get('LDS_Block_Size', '?'):>6s} vgpr={r.get('VGPR_Count', '?'):>4s} {name}", file=out)
    prev_end = e
span = (int(seg[-1]['End_Timestamp']) - t0) / 1e3
print(f"# launches {len(seg)}  span {span:.1f} us  kernel-busy {busy:.1f} us", file=out)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {n:42s} n={c:3d} {t:8.1f} us  avg {t / c:6.2f}", file=out)
