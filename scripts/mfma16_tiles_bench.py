#!/usr/bin/env python
"""Every tile of the 16 x 16 x 32 set against its 32 x 32 x 16 twin on the decoder's down / o shapes at S = 1621 (explicit variants, interleaved)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, M, N, K in (("down", 1621, 4096, 14336), ("o", 1621, 4096, 4096)):
    a = torch.randn((M, K), device=dev, generator=g).bfloat16()
    w = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16()
    res = torch.randn((M, N), device=dev, generator=g).bfloat16()
    st = torch.zeros((M, N // 64, 2), device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    cases = [(224, False), (224, True), (192, False), (192, True), (256, False), (256, True), (8, False), (16, True), (26, True), (4, False), (0, False), (0, True)]
    t = {c: [] for c in cases}
    for r in range(rounds):
        for c in cases:
            ops.set_gemm_variant(c[0])
            if r == 0: print('case', name, c, flush=True); torch.cuda.synchronize()
            for _ in range(2):
                ops.gemm(a, w, res=res, stats_out=st, out=out, mfma16=c[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, res=res, stats_out=st, out=out, mfma16=c[1])
            e1.record()
            torch.cuda.synchronize()
            t[c].append(e0.elapsed_time(e1) * 1e3 / 10)
    ops.set_gemm_variant(0)
    print(name, " | ".join(f"v{c[0]}{'+16' if c[1] else ''}: {sorted(t[c])[len(t[c]) // 2]:.1f}" for c in cases))
