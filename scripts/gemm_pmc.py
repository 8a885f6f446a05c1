#!/usr/bin/env python
"""Tiny driver for rocprofv3 --pmc passes: a few launches of each GEMM variant on one big shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
M, N, K = 9216, 4096, 4096
a = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for v in (1, 4):
    ops.set_gemm_variant(v)
    for _ in range(3):
        ops.gemm(a, w, out=c)
torch.cuda.synchronize()
