#!/usr/bin/env python
"""A/B: attention variant 3 (shipped) vs variant 4 (softmax denominators on the matrix pipe), interleaved, with the difference of the outputs."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videollama2_amd import ops
from scripts.kernel_bench import rnd, timeit
VARS = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,4").split(","))

def ab(fn, out):
    best, outs = {}, {}
    for _ in range(4):
        for v in VARS:
            ops.set_attn_kv_groups(v)
            best[v] = min(best.get(v, 1e9), timeit(fn, iters=30))
            outs[v] = out.float().clone()
    ops.set_attn_kv_groups(0)
    d = (outs[VARS[0]] - outs[VARS[1]]).norm() / outs[VARS[0]].norm()
    return best, float(d)

for B in (16, 8, 32):
    H, N, D = 16, 577, 64
    qkv = rnd(B * N, 3 * H * D)
    o = torch.empty(B * N, H * D, dtype=torch.bfloat16, device="cuda")
    st = (N * 3 * H * D, D, 3 * H * D)
    best, d = ab(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D), o)
    print(json.dumps(dict(shape=f"vit T={B}", **{f"v{v}_us": round(best[v], 1) for v in VARS}, rel_diff=d)), flush=True)
D, smax = 128, 4096
for name, S, nh, nkv in (("T8 7B", 945, 32, 8), ("T16 7B", 1621, 32, 8), ("T32 7B", 2973, 32, 8)):
    q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
    o = torch.empty(S, nh * D, dtype=torch.bfloat16, device="cuda")
    best, d = ab(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D), o)
    print(json.dumps(dict(shape=f"causal {name} S={S}", **{f"v{v}_us": round(best[v], 1) for v in VARS}, rel_diff=d)), flush=True)
