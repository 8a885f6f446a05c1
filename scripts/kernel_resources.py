#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table of libvl2hip.so (hipcc -Rpass-analysis=kernel-resource-usage): the check to
run after every kernel edit -- a kernel that starts spilling (ScratchSize > 0) or drops a wave of occupancy shows up here, on
the CPU-only build box, before any GPU time is spent.   python scripts/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "videollama2_amd", "csrc", "vl2_abi.hip")


def main():
    filt = sys.argv[1:]
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-finite-math-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "/dev/null"], capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(1)
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = dict(name=subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0])
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
    for k in rows:
        if filt and not any(f in k["name"] for f in filt):
            continue
        print(f"{k['name'][:78]:78s} {k.get('VGPRs', 0):5d} {k.get('AGPRs', 0):5d} {k.get('SGPRs', 0):5d} {k.get('ScratchSize [bytes/lane]', 0):8d} "
              f"{k.get('Occupancy [waves/SIMD]', 0):4d} {k.get('LDS Size [bytes/block]', 0):7d}")
    bad = [k["name"] for k in rows if k.get("ScratchSize [bytes/lane]", 0) > 0]
    print(f"{len(rows)} kernels; spilling: {bad if bad else 'none'}")


if __name__ == "__main__":
    main()
