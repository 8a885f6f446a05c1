#!/usr/bin/env python
"""GEMM time vs K at fixed M,N (fits t = a + b*K: a = per-launch + per-tile prologue/epilogue cost)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
from scripts.kernel_bench import timeit, rnd
for (M, N) in ((9232, 4096), (9232, 1024), (1621, 4096)):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda"); res = rnd(M, N)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        t_plain = timeit(lambda: ops.gemm(a, w, out=c))
        t_full = timeit(lambda: ops.gemm(a, w, bias=bias, res=res, act=1, out=c))
        print(f"M={M} N={N} K={K:5d}  plain {t_plain:7.1f} us  bias+qgelu+res {t_full:7.1f} us   ({2.0*M*N*K/t_plain/1e6:6.0f} TF/s)", flush=True)
