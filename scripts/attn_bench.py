#!/usr/bin/env python
"""A/B of the causal D=128 prefill attention with one / two KV groups per workgroup (VL2_TUNE_ATTN_KV_GROUPS), interleaved rounds,
at the prefill lengths of the T=8/16/32 workloads, the 72B head count and a 4-sequence batch.  Usage: python scripts/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

D, smax = 128, 4096
for name, S, nh, nkv, B in (("T8 7B", 945, 32, 8, 1), ("T16 7B", 1621, 32, 8, 1), ("T32 7B", 2973, 32, 8, 1), ("T16 v21", 1452, 28, 4, 1),
                            ("T16 72B", 1621, 64, 8, 1), ("T16 7B x4", 1621, 32, 8, 4)):
    q, kc, vc = rnd(B, S, nh * D), rnd(B, nkv, smax, D), rnd(B, nkv, smax, D)
    o = torch.empty(B, S, nh * D, dtype=torch.bfloat16, device="cuda")
    best = {}
    for r in range(3):
        for g in (1, 2):
            ops.set_attn_kv_groups(g)
            us = timeit(lambda: ops.attn_fwd(q, kc, vc, o, (S * nh * D, D, nh * D), (nkv * smax * D, smax * D, D), (nkv * smax * D, smax * D, D),
                                             (S * nh * D, D, nh * D), B, nh, S, S, nh // nkv, D ** -0.5, True, 0, D))
            best[g] = min(best.get(g, 1e9), us)
    ops.set_attn_kv_groups(0)
    fl = 4.0 * B * nh * (S * (S + 1) / 2) * D
    print(name, f"S={S} heads={nh} B={B}", {f"groups{g}": dict(us=round(u, 1), tflops=round(fl / u / 1e6, 1)) for g, u in best.items()}, flush=True)
