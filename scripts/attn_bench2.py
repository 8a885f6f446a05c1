#!/usr/bin/env python
"""A/B of the attention structures (vl2_attn_fwd variant 1/2 = k_attn.h register-staged, 3 = k_attn2.h LDS-DMA ring +
transpose reads, 4 = the same with two key streams per query block) at the workload's shapes, interleaved rounds in one process (guide rule 24), random data (rule 25).
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
        best = ab(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D), (0, 3, 4))
        fl = 4.0 * B * H * N * N * D
        print(json.dumps(dict(shape=f"vit T={B} 16x577x64", **{f"v{v}": dict(us=round(u, 1), tflops=round(fl / u / 1e6, 1)) for v, u in best.items()})), flush=True)
    D, smax = 128, 4096
    for name, S, nh, nkv in (("T8 7B", 945, 32, 8), ("T16 7B", 1621, 32, 8), ("T32 7B", 2973, 32, 8), ("T16 v21", 1452, 28, 4), ("T16 72B", 1621, 64, 8)):
        q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
        o = torch.empty(S, nh * D, dtype=torch.bfloat16, device="cuda")
        best = ab(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv,
                                       D ** -0.5, True, 0, D), (1, 2, 3, 4))
        fl = 4.0 * nh * (S * (S + 1) / 2) * D
        print(json.dumps(dict(shape=f"causal {name} S={S} heads={nh}/{nkv}", **{f"v{v}": dict(us=round(u, 1), tflops=round(fl / u / 1e6, 1)) for v, u in best.items()})), flush=True)


def extra():
    """Shapes around the chooser's rule for variant 4 (two key streams): short prompts, a chunk of new rows against a long cache, a batch."""
    D, smax, nh, nkv = 128, 4096, 32, 8
    for name, nq, nk, off, B in (("S=256", 256, 256, 0, 1), ("S=512", 512, 512, 0, 1), ("S=1152", 1152, 1152, 0, 1), ("S=1280", 1280, 1280, 0, 1), ("S=1408", 1408, 1408, 0, 1),
                                 ("chunk 128 @ 1621", 128, 1621, 1493, 1), ("chunk 512 @ 1621", 512, 1621, 1109, 1), ("chunk 512 @ 4096", 512, 4096, 3584, 1),
                                 ("batch 2 x S=945", 945, 945, 0, 2), ("batch 4 x S=945", 945, 945, 0, 4)):
        q, kc, vc = rnd(B, nq, nh * D), rnd(B, nkv, smax, D), rnd(B, nkv, smax, D)
        o = torch.empty(B, nq, nh * D, dtype=torch.bfloat16, device="cuda")
        best = ab(lambda: ops.attn_fwd(q, kc, vc, o, (nq * nh * D, D, nh * D), (nkv * smax * D, smax * D, D), (nkv * smax * D, smax * D, D), (nq * nh * D, D, nh * D),
                                       B, nh, nq, nk, nh // nkv, D ** -0.5, True, off, D), (3, 4))
        print(json.dumps(dict(shape=f"causal {name} heads={nh}/{nkv}", wgs=B * nh * ((nq + 127) // 128), **{f"v{v}": dict(us=round(u, 1)) for v, u in best.items()})), flush=True)


def skip_ab():
    """One-stream causal kernel with / without the skip of tiles the mask hides from a whole wave (lab variant 5 = without), libvl2hip_lab.so."""
    from videollama2_amd import _lib
    _lib.set_lab(True)
    D, smax = 128, 4096
    for name, S, nh, nkv in (("T16 7B", 1621, 32, 8), ("T32 7B", 2973, 32, 8), ("T16 72B", 1621, 64, 8), ("S=1792", 1792, 32, 8), ("S=2048", 2048, 32, 8),
                             ("S=2304", 2304, 32, 8), ("S=4096", 4096, 32, 8), ("T16 v21", 1452, 28, 4)):
        q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
        outs = {}
        for v in (3, 5):
            ops.set_attn_kv_groups(v)
            outs[v] = torch.zeros(S, nh * D, dtype=torch.bfloat16, device="cuda")
            ops.attn_fwd(q, kc, vc, outs[v], (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D)
        o = outs[3]
        best = ab(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv,
                                       D ** -0.5, True, 0, D), (3, 5), rounds=6)
        print(json.dumps(dict(shape=f"causal {name} S={S} heads={nh}/{nkv}", skip_us=round(best[3], 1), no_skip_us=round(best[5], 1),
                              same_bits=bool(torch.equal(outs[3], outs[5])))), flush=True)


def vit_order_ab():
    """The tower's class-token kernel with the query block as the slowest grid index (default since round 6) against the old order (lab variant 5)."""
    from videollama2_amd import _lib
    _lib.set_lab(True)
    for B in (16, 8, 32, 4):
        H, N, D = 16, 577, 64
        qkv = rnd(B * N, 3 * H * D)
        st = (N * 3 * H * D, D, 3 * H * D)
        outs = {}
        for v in (0, 5):
            ops.set_attn_kv_groups(v)
            outs[v] = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device="cuda")
            ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], outs[v], st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
        o = outs[0]
        best = ab(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D), (0, 5), rounds=6)
        print(json.dumps(dict(shape=f"vit T={B} 16x577x64", block_slowest_us=round(best[0], 1), old_order_us=round(best[5], 1), same_bits=bool(torch.equal(outs[0], outs[5])))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "vit_order":
        vit_order_ab()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "skip":
        skip_ab()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
        sys.exit(0)
    main()
