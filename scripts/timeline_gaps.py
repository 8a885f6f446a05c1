#!/usr/bin/env python
"""Gap analysis of a rocprofv3 --kernel-trace CSV: for the LAST timed forward step of bench.py (patchify ... argmax), the time
spent inside kernels vs between them, per kernel family.  usage: timeline_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "patchify" in n.split("(")[0]]
i0 = starts[-2] if len(starts) >= 2 else starts[-1]     # the last TIMED step (the profiled extra pass follows it)
i1 = starts[-1] if len(starts) >= 2 else len(rows)
seg = rows[i0:i1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step: {len(seg)} kernels, wall {1e-6*(t1-t0):.3f} ms, in-kernel {1e-6*busy:.3f} ms, gaps {1e-6*(t1-t0-busy):.3f} ms")
fam = collections.OrderedDict()
for a, b in zip(seg, seg[1:] + [None]):
    n = a["Kernel_Name"].split("(")[0][:70]
    d = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) if b else 0
    f = fam.setdefault(n, [0, 0, 0])
    f[0] += 1; f[1] += d; f[2] += g
for n, (c, d, g) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:72s} n={c:4d} kern={1e-3*d:9.1f} us avg={1e-3*d/c:7.1f} gap_after_sum={1e-3*g:8.1f} avg_gap={1e-3*g/c:6.2f}")
big = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), a["Kernel_Name"][:50], b["Kernel_Name"][:50]) for a, b in zip(seg, seg[1:])), reverse=True)[:12]
for g, a, b in big:
    print(f"gap {1e-3*g:8.1f} us  after {a}  before {b}")
