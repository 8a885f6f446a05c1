#!/usr/bin/env python
"""Driver for rocprofv3 --pmc passes over the two attention kernels at the workload's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
B, H, Nn, D = 16, 16, 577, 64
qkv = rnd(B * Nn, 3 * H * D); o = torch.empty(B * Nn, H * D, dtype=torch.bfloat16, device=dev)
st = (Nn * 3 * H * D, D, 3 * H * D)
S, nh, nkv, D2, smax = 1621, 32, 8, 128, 4096
q, kc, vc = rnd(S, nh * D2), rnd(nkv, smax, D2), rnd(nkv, smax, D2)
o2 = torch.empty(S, nh * D2, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (Nn * H * D, D, H * D), B, H, Nn, Nn, 1, D ** -0.5, False, 0, D)
    ops.attn_fwd(q, kc, vc, o2, (0, D2, nh * D2), (0, smax * D2, D2), (0, smax * D2, D2), (0, D2, nh * D2), 1, nh, S, S, nh // nkv, D2 ** -0.5, True, 0, D2)
torch.cuda.synchronize()
