#!/usr/bin/env python
"""The decoder's o / down projections (residual + row statistics) on the two matrix instructions, interleaved on one box through the C ABI: the family's
automatic choice (v_mfma_f32_32x32x16: fill-the-round / 128x256 / mixed tiles) against the 16 x 16 x 32 set's (VL2_GEMM_MFMA16: gemm7_16 / the one-round
128 x 128 body / gemm9 + mixed launch).  Usage: python scripts/mfma16_set_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, N, K in (("down", 4096, 14336), ("o", 4096, 4096), ("q/k/v (no residual)", 6144, 4096)):
        w = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16()
        for M in (945, 1621, 2973):
            a = torch.randn((M, K), device=dev, generator=g).bfloat16()
            res = torch.randn((M, N), device=dev, generator=g).bfloat16() if N == 4096 else None
            st = torch.zeros((M, N // 64, 2), device=dev) if N == 4096 else None
            out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            t = {False: [], True: []}
            for r in range(rounds):
                for mf in (False, True):
                    for _ in range(3):
                        ops.gemm(a, w, res=res, stats_out=st, out=out, mfma16=mf)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ops.gemm(a, w, res=res, stats_out=st, out=out, mfma16=mf)
                    e1.record()
                    torch.cuda.synchronize()
                    t[mf].append(e0.elapsed_time(e1) * 1e3 / 20)
            fl = 2.0 * M * N * K
            m0, m1 = sorted(t[False])[rounds // 2], sorted(t[True])[rounds // 2]
            print(f"{name:22s} M={M:5d}: 32x32x16 family {m0:7.1f} us ({fl / m0 * 1e-6:6.1f} TF/s)   16x16x32 set {m1:7.1f} us ({fl / m1 * 1e-6:6.1f} TF/s)   ratio {m1 / m0:.3f}")


if __name__ == "__main__":
    main()
