#!/usr/bin/env python
"""Micro-benchmark of the fill-the-round GEMM tiles (csrc/k_gemm7.h, variants 224 / 192) against the other tile shapes on the one-round
shapes of the T = 16 step, through the C ABI, variants INTERLEAVED in one process on one box.  Usage: python scripts/gemm7_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from videollama2_amd.connector import conv3d_k2s2p1_index  # noqa: E402

dev = "cuda"


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.attach_workspace(dev)
    cases = []
    for name, M, N, K, kw in (("llm_down", 1621, 4096, 14336, dict(res=True, stats=True)), ("llm_wo", 1621, 4096, 4096, dict(res=True, stats=True)),
                              ("stc_s2_conv", 1521, 4096, 4096, dict()), ("stc_s2_ro0", 1521, 4096, 4096, dict(bias=True, act=2)),
                              ("llm_wo_T8", 945, 4096, 4096, dict(res=True, stats=True)), ("llm_down_T8", 945, 4096, 14336, dict(res=True, stats=True)),
                              ("llm_wo_T12", 1283, 4096, 4096, dict(res=True, stats=True)), ("llm_wo_T24", 2297, 4096, 4096, dict(res=True, stats=True)),
                              ("72b_wo", 1621, 8192, 8192, dict(res=True, stats=True))):
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        res = rnd(M, N) if kw.get("res") else None
        so = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=dev) if kw.get("stats") else None
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        cases.append((name, M, N, K, lambda a=a, w=w, bias=bias, res=res, so=so, out=out, act=kw.get("act", 0): ops.gemm(a, w, bias=bias, res=res, act=act, stats_out=so, out=out)))
    T, H, C = 16, 24, 4096
    x, wp, b = rnd(T * H * H, C), rnd(C, 8 * C, scale=(8 * C) ** -0.5), torch.randn(C, device=dev)
    idx, _ = conv3d_k2s2p1_index(T, H, H, dev)
    zero = torch.zeros(C, dtype=torch.bfloat16, device=dev)
    cases.append(("stc_conv3d_gather", 1521, 4096, 8 * C, lambda: ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C))))
    variants = (0, 1, 4, 5, 8, 12, 192, 193, 224, 225)
    tab = {c[0]: {v: [] for v in variants} for c in cases}
    for _ in range(rounds):
        for name, M, N, K, fn in cases:
            for v in variants:
                if name == "stc_conv3d_gather" and v in (4, 5, 8, 12, 193, 225):
                    continue
                ops.set_gemm_variant(v)
                try:
                    tab[name][v].append(timeit(fn))
                finally:
                    ops.set_gemm_variant(0)
    for name, M, N, K, fn in cases:
        fl = 2.0 * M * N * K
        print(f"{name:18s} {M}x{N}x{K}: " + "  ".join(f"v{v}: " + "/".join(f"{t:.1f}" for t in ts) + f" us ({fl / min(ts) / 1e6:.0f} TF/s)" for v, ts in tab[name].items() if ts))


if __name__ == "__main__":
    main()
