#!/usr/bin/env python
"""Merge the three rocprofv3 --pmc passes of scripts/gpu_gemm_pmc.sh over scripts/gemm_pmc.py into one row per (shape, kernel): the
THIRD launch of every shape (each shape is launched three times; a row-split call is two kernels).  Adds MFMA-busy share =
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs (both counters are summed over the chip)."""
import csv, glob, sys
from collections import defaultdict, OrderedDict
sys.path.insert(0, "scripts")
from gemm_pmc import SHAPES
rows = OrderedDict()
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rs = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
    disp = OrderedDict()
    for r in rs:
        disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("(")[0].replace("void ", "")})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    # group consecutive dispatches by shape: every shape = 3 calls of the same kernel sequence
    seqs, i = [], 0
    names = [disp[j]["k"] for j in ids]
    for M, N, K, v, sw in SHAPES:
        # length of one call = number of dispatches until the pattern repeats (1 or 2)
        n = 2 if (i + 3 < len(names) and names[i] != names[i + 1] and names[i] == names[i + 2]) else 1
        for t in range(n):
            key = (f"{M}x{N}x{K}" + (" swiglu" if sw else "") + f" (variant {v})", names[i + 2 * n + t])
            rows.setdefault(key, {}).update({c: x for c, x in disp[ids[i + 2 * n + t]].items() if c != "k"})
        i += 3 * n
cols = sorted({c for r in rows.values() for c in r})
w = csv.writer(sys.stdout)
w.writerow(["shape", "kernel"] + cols + ["mfma_busy_share"])
for (shape, k), r in rows.items():
    busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (r.get("GRBM_GUI_ACTIVE", 1) / 8) if r.get("GRBM_GUI_ACTIVE") else ""
    w.writerow([shape, k] + [int(r.get(c, 0)) for c in cols] + [round(busy, 3) if busy != "" else ""])
