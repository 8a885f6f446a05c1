#!/usr/bin/env python
"""Probe: ViT fc1 (9232 x 4096 x 1024, bias + QuickGELU) as whole rounds of the 256x256 kernel + a tail on another kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videollama2_amd import ops
dev = "cuda"
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K, act in (("vit_fc1", 9232, 4096, 1024, 1), ("vit_qkv", 9232, 3072, 1024, 0), ("stc_s1_b1", 9216, 4096, 1024, 0)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    def whole(v):
        ops.set_gemm_variant(v); ops.gemm(a, w, bias=bias, act=act, out=c)
    res = {f"whole v{v}": timeit(lambda: whole(v)) for v in (1, 8, 0)}
    ref = c.clone()
    per_round = 256 // (N // 256) if 256 % (N // 256) == 0 else None
    if per_round:
        for rounds in (1, 2):
            M1 = rounds * per_round * 256
            if M1 >= M: continue
            for vt in (1, 4, 8, 32):
                def split():
                    ops.set_gemm_variant(8); ops.gemm(a[:M1], w, bias=bias, act=act, out=c[:M1])
                    ops.set_gemm_variant(vt); ops.gemm(a[M1:], w, bias=bias, act=act, out=c[M1:])
                res[f"{rounds} rounds v8 + tail({M - M1} rows) v{vt}"] = timeit(split)
                assert torch.equal(c, ref)
    ops.set_gemm_variant(0)
    print(name, {k: round(v, 1) for k, v in res.items()}, flush=True)
