#!/usr/bin/env python
"""gate/up + SwiGLU on the default mixed 16 x 16 x 32 launch with other depths of the row-tile group that walks one W panel together on an XCD (k_gemm9.h
gemm9_body: 4 shipped = variant 26; lab variants 27 / 29 = 8 / 6; 28 = depth 4 with the empty waves of the tail tiles computing, i.e. without the dead-wave skip), one binary (libvl2hip_lab.so), alternating.  Usage: python scripts/gemm9_group_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import _lib, ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    _lib.set_lab(True)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    K, N = 4096, 28672
    w = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16()
    for M in (1621, 2973, 945):
        a = torch.randn((M, K), device=dev, generator=g).bfloat16()
        rn = ops.row_norm_finalize(ops.row_stats(a), K, ops.NORM_RMS, 1e-5)
        vs = (26, 27, 28, 29)
        outs = {v: torch.empty((M, N // 2), device=dev, dtype=torch.bfloat16) for v in vs}
        res = {v: [] for v in vs}
        for r in range(rounds + 1):
            for v in vs:
                ops.set_gemm_variant(v)
                for _ in range(3):
                    ops.gemm(a, w, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-5, None), out=outs[v])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.gemm(a, w, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-5, None), out=outs[v])
                e1.record()
                torch.cuda.synchronize()
                if r:                                                   # (round 0 warms the clocks)
                    res[v].append(e0.elapsed_time(e1) * 1e3 / 20)
        ops.set_gemm_variant(0)
        fl = 2.0 * M * N * K
        for v, depth in zip(vs, (4, 8, '4, no dead-wave skip', 6)):
            t = sorted(res[v])[len(res[v]) // 2]
            print(f"gate/up M={M}: group depth {depth} (variant {v}) median {t:7.1f} us  {fl / t * 1e-6:7.1f} TF/s   all {[round(x, 1) for x in res[v]]}   same bits {bool(torch.equal(outs[v], outs[26]))}")


if __name__ == "__main__":
    main()
