#!/usr/bin/env python
"""Probe: does the ViT attention kernel pay for partially filled rounds of resident workgroups?  Time vs the number of frames B
(workgroups = 5 * 16 * B; four workgroups of four waves fit a CU at 114 VGPRs -> 1024 resident): a staircase means the tail round
matters, a straight line means it does not."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videollama2_amd import ops
from scripts.kernel_bench import rnd, timeit

H, N, D = 16, 577, 64
res = {}
for rnd_i in range(3):
    for B in (8, 10, 12, 13, 14, 16, 19, 20, 24, 25, 26, 32):
        qkv = rnd(B * N, 3 * H * D)
        o = torch.empty(B * N, H * D, dtype=torch.bfloat16, device="cuda")
        st = (N * 3 * H * D, D, 3 * H * D)
        us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D), iters=30)
        res[B] = min(res.get(B, 1e9), us)
for B, us in sorted(res.items()):
    print(json.dumps(dict(frames=B, workgroups=5 * H * B, rounds_of_1024=round(5 * H * B / 1024, 2), us=round(us, 1), us_per_frame=round(us / B, 2))), flush=True)
