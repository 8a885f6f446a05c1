#!/usr/bin/env python
"""The 160-row tile (variant 10, round 6) against the 192- and 256-row tiles on the tower's N = 1024 shapes (out_proj, fc2 with residual + row
statistics), interleaved on one box through the C ABI.  Usage: python scripts/tile160_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K in (("ViT out_proj T=16", 9232, 1024, 1024), ("ViT fc2 T=16", 9232, 1024, 4096), ("ViT fc2 T=32", 18464, 1024, 4096),
                          ("ViT out_proj T=32", 18464, 1024, 1024), ("ViT fc2 T=8", 4616, 1024, 4096)):
        a = torch.randn((M, K), device=dev, generator=g).bfloat16()
        w = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16()
        res = torch.randn((M, N), device=dev, generator=g).bfloat16()
        bias = torch.randn((N,), device=dev, generator=g)
        st = torch.zeros((M, N // 64, 2), device=dev)
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        vs = (0, 8, 12, 10)
        t = {v: [] for v in vs}
        for r in range(rounds):
            for v in vs:
                ops.set_gemm_variant(v)
                for _ in range(3):
                    ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=out)
                e1.record()
                torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) * 1e3 / 20)
        ops.set_gemm_variant(0)
        fl = 2.0 * M * N * K
        print(name, f"{M}x{N}x{K}:", "  ".join(f"variant {v}: {sorted(t[v])[len(t[v]) // 2]:6.1f} us ({fl / sorted(t[v])[len(t[v]) // 2] * 1e-6:6.1f} TF/s)" for v in vs))


if __name__ == "__main__":
    main()
