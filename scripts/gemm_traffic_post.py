#!/usr/bin/env python
"""Post-process the FETCH_SIZE / WRITE_SIZE passes of scripts/gemm_traffic_pmc.py into per-launch HBM traffic.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2;
WRITE_SIZE is used as reported.  Both counters are in KiB."""
import csv, json, sys
COUNTS = [1, 23, 23, 23, 23, 2, 7, 10, 32, 32, 32, 32, 1]     # launches per step of each shape, in driver order
ALG = None
def per_shape(path, counter):
    """One value per vl2_gemm CALL: the kernels between two separator fills (scripts/gemm_traffic_pmc.py) are summed -- a row-split
    call is two GEMM kernels back to back.  Calls come in pairs (warm-up, measured): the second of each pair is kept."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    calls, cur, names = [], 0.0, []
    for r in rows:
        if "gemm" in r["Kernel_Name"]:
            cur += float(r["Counter_Value"])
            names.append(r["Kernel_Name"].split("(")[0].replace("void ", ""))
        elif "fill" in r["Kernel_Name"].lower() and names:
            calls.append((cur, " + ".join(names)))
            cur, names = 0.0, []
    assert len(calls) == 2 * len(COUNTS), (len(calls), counter)
    return [c[0] for c in calls[1::2]], [c[1] for c in calls[1::2]]
fetch, names = per_shape(sys.argv[1], "FETCH_SIZE")
write, _ = per_shape(sys.argv[2], "WRITE_SIZE")
tot = sum(c * (2 * f + w) * 1024 for c, f, w in zip(COUNTS, fetch, write))
n = sum(COUNTS)
out = dict(source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/gemm_traffic_pmc.py; FETCH x2 (gfx950)",
           launches_per_step=n, hbm_bytes_per_launch=round(tot / n), hbm_bytes_per_step=round(tot),
           per_shape=[dict(kernel=k, launches=c, fetch_kib=f, write_kib=w) for k, c, f, w in zip(names, COUNTS, fetch, write)])
print(json.dumps(out, indent=1))
