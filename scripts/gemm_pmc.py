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
SHAPES = [                                   # (M, N, K, variant, swiglu): the kernels HEAD dispatches on the step's dominant shapes
    (8192, 4096, 4096, 8, False),            # 256x256 ping-pong, C^T epilogue, well quantised
    (9232, 3072, 1024, 0, False),            # ViT qkv (auto: 256x256, C^T epilogue)
    (9232, 1024, 4096, 12, False),           # ViT fc2 shape on 192-row tiles (no residual here: C^T epilogue)
    (1621, 6144, 4096, 0, False),            # LLM q/k/v (auto: 192-row tiles)
    (1621, 28672, 4096, 0, True),            # gate/up + SwiGLU (auto: 256x256 on 1536 rows + one-round kernel on 85)
    (1621, 4096, 14336, 0, False),           # down (auto: 128x256 ping-pong)
    (9232, 4096, 1024, 0, False),            # ViT fc1 shape (auto)
]
if __name__ == "__main__":
    for M, N, K, v, sw in SHAPES:
        go(M, N, K, v, sw)
    ops.set_gemm_variant(0)
