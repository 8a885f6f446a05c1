#!/usr/bin/env python
"""The STC stage-s1 convolutions (N = 4096, K = 4096 / 1024) at every row count their call sites see (576 rows per frame: 1 ... 32 frames) on the family's choice
and on the 16 x 16 x 32 set (VL2_GEMM_MFMA16), alternating.  Usage: python scripts/s1_mfma16_probe.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ops.attach_workspace("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    for K in (4096, 1024):
        w = (torch.randn((4096, K), device="cuda", generator=g) * K ** -0.5).bfloat16()
        for T in (1, 2, 4, 8, 16, 32):
            M = 576 * T
            a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
            out = torch.empty((M, 4096), device="cuda", dtype=torch.bfloat16)
            res = {False: [], True: []}
            for r in range(rounds + 1):
                for mf in (False, True):
                    for _ in range(3):
                        ops.gemm(a, w, out=out, mfma16=mf)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ops.gemm(a, w, out=out, mfma16=mf)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        res[mf].append(e0.elapsed_time(e1) * 1e3 / 20)
            med = lambda x: sorted(x)[len(x) // 2]
            print(f"s1 conv {M:6d} x 4096 x {K}: family {med(res[False]):7.1f} us   16x16x32 set {med(res[True]):7.1f} us   ({100 * (med(res[True]) / med(res[False]) - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
