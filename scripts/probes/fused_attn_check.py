#!/usr/bin/env python
"""Probe: vl2_attn_decode_fused vs vl2_attn_decode on random data, several context lengths / head layouts; prints where they differ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videollama2_amd import ops
dev = "cuda" if torch.cuda.is_available() else "cpu"
D = 128
for nh, nkv, smax, pos in ((32, 8, 2048, 300), (32, 8, 2048, 1650), (28, 4, 512, 300), (28, 4, 2048, 300), (32, 8, 512, 300), (32, 8, 2048, 63), (32, 8, 2048, 64)):
    g = torch.Generator().manual_seed(nh + pos)
    qkv = (torch.randn((nh + 2 * nkv) * D, generator=g)).bfloat16().to(dev)
    kc = torch.randn(nkv, smax, D, generator=g).bfloat16().to(dev); vc = torch.randn(nkv, smax, D, generator=g).bfloat16().to(dev)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D)); fr = torch.arange(smax).float()[:, None] * inv[None]
    cos_t, sin_t = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
    nsp = (smax + 63) // 64
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=dev)
    res = []
    for rep in range(3):
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        p1 = torch.full((nh * nsp * 130,), 7.0, device=dev); p2 = torch.full((nh * nsp * 130,), -3.0, device=dev)
        o1 = torch.zeros(nh * D, dtype=torch.bfloat16, device=dev); o2 = torch.zeros_like(o1)
        ops.attn_decode(qkv, k1, v1, cos_t, sin_t, p1, o1, nh, nkv, pos, D ** -0.5, pos_dev=pos_dev, ctx_cap=smax)
        cnt = torch.zeros(nkv, dtype=torch.int32, device=dev)
        ops.attn_decode_fused(qkv, k2, v2, cos_t, sin_t, p2, o2, nh, nkv, pos_dev, D ** -0.5, cnt)
        if dev == "cuda": torch.cuda.synchronize()
        bad = (o1 != o2).nonzero().flatten()
        res.append((int(bad.numel()), sorted(set((bad // D).tolist()))[:8], bool(torch.equal(k1, k2) and torch.equal(v1, v2)), cnt.tolist()))
    print(nh, nkv, smax, pos, res, flush=True)
