#!/usr/bin/env python
"""Decode GEMVs at the VideoLLaMA2-7B shapes on 16-bit weights (vl2_gemv_bf16) and on their fp8 copies (vl2_gemv_fp8, csrc/k_fp8.h), eager
launches back to back (the small ones are launch-bound here; inside the decode graph they are not).  Usage: python scripts/fp8_bench.py"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops
from scripts.kernel_bench import timeit, rnd
dev = "cuda"
out = {}
for name, N, K, kw in (("qkv", 6144, 4096, dict(rms=True)), ("o", 4096, 4096, dict()), ("gate_up", 28672, 4096, dict(rms=True, swiglu=True)),
                       ("down", 4096, 14336, dict()), ("lm_head", 32000, 4096, dict(f32=True))):
    w, x = rnd(N, K, scale=K ** -0.5), rnd(K)
    q, sc = ops.quant_fp8(w)
    ones = torch.ones(K, device=dev)
    u16 = timeit(lambda: ops.gemv(w, x, norm_w=ones if kw.get("rms") else None, swiglu=kw.get("swiglu", False), out_f32=kw.get("f32", False)), iters=100)
    u8 = timeit(lambda: ops.gemv_fp8(q, sc, x, swiglu=kw.get("swiglu", False), out_f32=kw.get("f32", False), rms_plain=kw.get("rms", False)), iters=100)
    out[name] = dict(us_16bit=round(u16, 1), tbs_16bit=round(N * K * 2 / u16 / 1e6, 2), us_fp8=round(u8, 1), tbs_fp8=round(N * K / u8 / 1e6, 2))
    print(name, out[name], flush=True)
print("JSON", json.dumps(out))
