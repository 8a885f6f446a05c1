#!/usr/bin/env python
"""Tiny driver for rocprofv3 --pmc passes: a few launches of each GEMM kernel on the shapes that dominate the forward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
def go(M, N, K, v, swiglu=False):
    a = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N // 2 if swiglu else N, dtype=torch.bfloat16, device=dev)
    ops.set_gemm_variant(v)
    for _ in range(3):
        ops.gemm(a, w, swiglu=swiglu, out=c)
    torch.cuda.synchronize()
go(8192, 4096, 4096, 8)            # 256x256 ping-pong, well quantised
go(9216, 4096, 4096, 4)            # 128x256 ping-pong, STC s1
go(1621, 28672, 4096, 4, True)     # 128x256 ping-pong, gate-up + SwiGLU
go(1621, 4096, 14336, 4)           # down
go(9232, 4096, 1024, 1)            # 128x128, ViT fc1 (K=1024)
