#!/usr/bin/env python
"""A/B of the two attention structures (vl2_attn_fwd variant 1/2 = k_attn.h register-staged, 3 = k_attn2.h LDS-DMA ring +
transpose reads) at the workload's shapes, interleaved rounds in one process (guide rule 24), random data (rule 25).
Prints one JSON line per shape.  Usage: python scripts/attn_bench2.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402


def ab(fn, variants, rounds=4):
    best = {}
    for _ in range(rounds):
        for v in variants:
            ops.set_attn_kv_groups(v)
            best[v] = min(best.get(v, 1e9), timeit(fn, iters=30))
    ops.set_attn_kv_groups(0)
    return best


def main():
    # ViT: B frames x 16 heads x 577 tokens x 64, straight out of the fused qkv buffer
    for B in (16, 8, 32):
        H, N, D = 16, 577, 64
        qkv = rnd(B * N, 3 * H * D)
        o = torch.empty(B * N, H * D, dtype=torch.bfloat16, device="cuda")
        st = (N * 3 * H * D, D, 3 * H * D)
        best = ab(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D), (1, 3))
        fl = 4.0 * B * H * N * N * D
        print(json.dumps(dict(shape=f"vit T={B} 16x577x64", **{f"v{v}": dict(us=round(u, 1), tflops=round(fl / u / 1e6, 1)) for v, u in best.items()})), flush=True)
    D, smax = 128, 4096
    for name, S, nh, nkv in (("T8 7B", 945, 32, 8), ("T16 7B", 1621, 32, 8), ("T32 7B", 2973, 32, 8), ("T16 v21", 1452, 28, 4), ("T16 72B", 1621, 64, 8)):
        q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
        o = torch.empty(S, nh * D, dtype=torch.bfloat16, device="cuda")
        best = ab(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv,
                                       D ** -0.5, True, 0, D), (1, 2, 3))
        fl = 4.0 * nh * (S * (S + 1) / 2) * D
        print(json.dumps(dict(shape=f"causal {name} S={S} heads={nh}/{nkv}", **{f"v{v}": dict(us=round(u, 1), tflops=round(fl / u / 1e6, 1)) for v, u in best.items()})), flush=True)


if __name__ == "__main__":
    main()
