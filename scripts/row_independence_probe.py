#!/usr/bin/env python
"""Diagnostic: does each per-frame / per-row op give the same bits on a subset of frames as on all of them?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
bf = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).bfloat16().to(dev)
ops.attach_workspace(dev)
C = 4096
# small_linear
for (N, K) in ((1024, 4096), (4096, 1024)):
    x = torch.randn(8, K, generator=g).to(dev); w = bf(N, K, scale=K ** -0.5); b = torch.randn(N, generator=g).to(dev)
    full = ops.small_linear(x, w, b, ops.ACT_SILU)
    for lo, hi in ((0, 2), (2, 4), (6, 8), (3, 8)):
        sub = ops.small_linear(x[lo:hi].contiguous(), w, b, ops.ACT_SILU)
        print("small_linear", N, K, lo, hi, torch.equal(sub, full[lo:hi]), (sub - full[lo:hi]).abs().max().item())
# gemm row subsets at the sharded shapes
for (M, Ms, N, K) in ((845, 338, 4096, 4096), (845, 169, 4096, 4096), (4616, 1154, 1024, 4096), (4616, 1154, 1024, 1024), (4616, 1154, 3072, 1024),
                      (4616, 1154, 4096, 1024), (4608, 1152, 4096, 4096), (4608, 1152, 4096, 1024)):
    a, w = bf(M, K), bf(N, K, scale=K ** -0.5)
    full = ops.gemm(a, w)
    sub = ops.gemm(a[M - Ms:].contiguous(), w)
    print("gemm", M, Ms, N, K, torch.equal(sub, full[M - Ms:]))
# chan_mean / dwconv / se_scale / layernorm on frame subsets
F, H = 5, 13
x = bf(F * H * H, C)
m = ops.chan_mean(x, F, H * H)
m2 = ops.chan_mean(x[3 * H * H:].contiguous(), 2, H * H)
print("chan_mean", torch.equal(m2, m[3:]))
w9, lw, lb = torch.randn(9, C, generator=g).to(dev), torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
d = ops.dwconv3x3_ln_silu(x, w9, lw, lb, F, H, H, 1e-5)
d2 = ops.dwconv3x3_ln_silu(x[3 * H * H:].contiguous(), w9, lw, lb, 2, H, H, 1e-5)
print("dwconv", torch.equal(d2, d[3 * H * H:]))
ln = ops.layernorm(x, lw, lb, 1e-5, silu=True)
ln2 = ops.layernorm(x[3 * H * H:].contiguous(), lw, lb, 1e-5, silu=True)
print("layernorm", torch.equal(ln2, ln[3 * H * H:]))
