#!/usr/bin/env python
"""Driver for a rocprofv3 --pmc FETCH_SIZE pass over the decode GEMVs of one Mistral-7B layer (+ lm_head): the bytes the
memory side delivers per launch should equal the weight bytes (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
dev = "cuda"
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x, nw = rnd(4096), torch.ones(4096, device=dev)
for name, N, K, kw in (("qkv", 6144, 4096, dict(norm_w=nw)), ("wo", 4096, 4096, dict()), ("gateup", 28672, 4096, dict(norm_w=nw, swiglu=True)),
                       ("down", 4096, 14336, dict()), ("lm_head", 32000, 4096, dict(norm_w=nw, out_f32=True))):
    w = rnd(N, K, scale=K ** -0.5)
    xv = rnd(K)
    for _ in range(3):
        ops.gemv(w, xv, **kw)
    torch.cuda.synchronize()
