#!/usr/bin/env python
"""Attention micro-benchmark at the workload's two shapes (ViT T = 16: 16 x 16 heads x 577 x 64, class token peeled; causal prefill
S = 1621: 32 / 8 heads x 128), 200 launches each.  For A/B runs of two library builds on one box (scripts/gpu_r4_w.sh swaps the .so)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
from scripts.kernel_bench import timeit, rnd
dev = "cuda"
out = {}
for T in (16, 32):
    B, H, Nn, D = T, 16, 577, 64
    qkv = rnd(B * Nn, 3 * H * D)
    o = torch.empty(B * Nn, H * D, dtype=torch.bfloat16, device=dev)
    st = (Nn * 3 * H * D, D, 3 * H * D)
    us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (Nn * H * D, D, H * D), B, H, Nn, Nn, 1, D ** -0.5, False, 0, D), iters=200)
    out[f"attn_vit_T{T}"] = round(us, 2)
for S in (1621, 2973):
    nh, nkv, D, smax = 32, 8, 128, 4096
    q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
    o = torch.empty(S, nh * D, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D), iters=200)
    out[f"attn_prefill_S{S}"] = round(us, 2)
print(json.dumps(out), flush=True)
