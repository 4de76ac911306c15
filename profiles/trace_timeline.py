#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 kernel trace of bench.py: kernel time vs gaps between the
big channelizer launches (to see how much of a step is launch/host overhead)."""
import csv, glob, sys
from collections import defaultdict
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], int(r.get("Grid_Size_X") or 0)))
rows.sort()
big = [i for i, r in enumerate(rows) if "k_channelize_mfma" in r[2] and r[3] == max(x[3] for x in rows if "k_channelize_mfma" in x[2])]
if len(big) < 2:
    sys.exit("need >= 2 steps")
a, b = big[-2], big[-1]
seg = rows[a:b]
span = seg[-1][1] if False else rows[b][0] - rows[a][0]
busy = sum(e - s for s, e, _, _ in seg)
print(f"step span {span/1e3:.1f} us, kernel-busy {busy/1e3:.1f} us, idle {100*(1-busy/span):.0f}%  ({len(seg)} launches)")
agg = defaultdict(lambda: [0, 0])
for s, e, n, g in seg:
    agg[n][0] += e - s; agg[n][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t/1e3:9.1f} us  x{c:<3d} {n}")
prev = seg[0][1]
gaps = []
for s, e, n, g in seg[1:]:
    gaps.append((s - prev, n)); prev = max(prev, e)
gaps.sort(reverse=True)
print("largest gaps (us) before kernel:", [(round(g/1e3, 1), n[:28]) for g, n in gaps[:8]])
