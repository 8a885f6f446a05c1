#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at the exact VideoLLaMA2-7B (T=16) shapes, through the C ABI.  A/B of the GEMM
tile variants in ONE process, interleaved rounds (guide rule 24).  Usage: python scripts/kernel_bench.py [--quick]"""
import os
import sys
import json

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


VARIANTS = (1, 4, 8)
SHAPES = [  # name, M, N, K, kwargs
    ("sq_8192x4096x4096", 8192, 4096, 4096, dict()),
    ("sq_8192", 8192, 8192, 8192, dict()),
    ("vit_qkv", 9232, 3072, 1024, dict(bias=True)),
    ("vit_wo", 9232, 1024, 1024, dict(bias=True, res=True)),
    ("vit_fc1", 9232, 4096, 1024, dict(bias=True, act=1)),
    ("vit_fc2", 9232, 1024, 4096, dict(bias=True, res=True)),
    ("stc_s1_conv", 9216, 4096, 4096, dict()),
    ("stc_s1_b1", 9216, 4096, 1024, dict()),
    ("stc_s2_conv", 1521, 4096, 4096, dict()),
    ("llm_qkv", 1621, 6144, 4096, dict()),
    ("llm_wo", 1621, 4096, 4096, dict(res=True)),
    ("llm_gateup", 1621, 28672, 4096, dict(swiglu=True)),
    ("llm_down", 1621, 4096, 14336, dict(res=True)),
]


SMALL = [  # per-rank shapes of the frame-sharded encoder at 8 ranks (2 / 4 frames per rank)
    ("vit_qkv_2f", 1154, 3072, 1024, dict(bias=True)), ("vit_wo_2f", 1154, 1024, 1024, dict(bias=True, res=True)),
    ("vit_fc1_2f", 1154, 4096, 1024, dict(bias=True, act=1)), ("vit_fc2_2f", 1154, 1024, 4096, dict(bias=True, res=True)),
    ("vit_qkv_4f", 2308, 3072, 1024, dict(bias=True)), ("vit_fc2_4f", 2308, 1024, 4096, dict(bias=True, res=True)),
    ("stc_s1_2f", 1152, 4096, 4096, dict()), ("stc_s1_b1_2f", 1152, 4096, 1024, dict()), ("stc_s1_4f", 2304, 4096, 4096, dict()),
    ("stc_s2_1to", 169, 4096, 4096, dict()), ("stc_s2_2to", 338, 4096, 4096, dict()),
    ("conv3d_as_plain_2to", 338, 4096, 32768, dict(bias=True, act=3)),
    ("vit_wo_4f", 2308, 1024, 1024, dict(bias=True, res=True)), ("vit_fc1_4f", 2308, 4096, 1024, dict(bias=True, act=1)),
    ("stc_s2_3to", 507, 4096, 4096, dict()), ("stc_s2_5to", 845, 4096, 4096, dict()), ("vit_fc2_8f", 4616, 1024, 4096, dict(bias=True, res=True)),
    ("vit_wo_8f", 4616, 1024, 1024, dict(bias=True, res=True)), ("vit_qkv_1f", 577, 3072, 1024, dict(bias=True)), ("vit_fc1_1f", 577, 4096, 1024, dict(bias=True, act=1)),
]


def shapes_for(T):
    """The GEMM shapes of a T-frame VideoLLaMA2-7B step (to check the per-shape kernel choice away from T=16)."""
    Mv, Ms, To = T * 577, T * 576, (T // 2 + 1) * 169
    S = To + 100
    return [("vit_qkv", Mv, 3072, 1024, dict(bias=True)), ("vit_wo", Mv, 1024, 1024, dict(bias=True, res=True)),
            ("vit_fc1", Mv, 4096, 1024, dict(bias=True, act=1)), ("vit_fc2", Mv, 1024, 4096, dict(bias=True, res=True)),
            ("stc_s1_conv", Ms, 4096, 4096, dict()), ("stc_s1_b1", Ms, 4096, 1024, dict()), ("stc_s2_conv", To, 4096, 4096, dict()),
            ("llm_qkv", S, 6144, 4096, dict()), ("llm_wo", S, 4096, 4096, dict(res=True)),
            ("llm_gateup", S, 28672, 4096, dict(swiglu=True)), ("llm_down", S, 4096, 14336, dict(res=True))]


def main():
    global SHAPES, VARIANTS
    out = {}
    ops.attach_workspace(dev)
    if "--frames" in sys.argv:
        SHAPES, VARIANTS = shapes_for(int(sys.argv[sys.argv.index("--frames") + 1])), (1, 0, 24, 8, 12, 60, 61)
        if "--peeled" in sys.argv:                      # the ViT GEMMs on the patch rows only (CLS rows peeled off: M = 576 T)
            T = int(sys.argv[sys.argv.index("--frames") + 1])
            SHAPES = [(n + "_p", T * 576, N, K, kw) for n, M, N, K, kw in SHAPES if n.startswith("vit_")] + SHAPES
    if "--small" in sys.argv:
        SHAPES, VARIANTS = SMALL, (1, 0, 32, 256)   # 0 = auto, 1 = 128x128, 32 = 64x64 small-M kernel, 's' = 128x128 + split-K
    rounds = 2 if "--quick" in sys.argv else 3
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20   # --iters 400: power steady state per variant
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
    if only:
        SHAPES = [s for s in SHAPES if s[0] in only]
    for name, M, N, K, kw in SHAPES:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        ncol = N // 2 if kw.get("swiglu") else N
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        res = rnd(M, ncol) if kw.get("res") else None
        c = torch.empty(M, ncol, dtype=torch.bfloat16, device=dev)
        best = {}
        for r in range(rounds):
            for v in VARIANTS:
                if v in (4, 8, 12, 60, 61) and N % 256:
                    continue
                ops.set_gemm_variant(1 if v == 's' else v)
                ops.set_splitk(v == 's')
                us = timeit(lambda: ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0), swiglu=kw.get("swiglu", False), out=c), iters=iters)
                best[v] = min(best.get(v, 1e9), us)
                if v == 1:
                    ref = c.clone()
                elif r == 0 and v not in ('s', 0):
                    d = (ref.float() - c.float()).abs().max().item()
                    if d > 0.05 * ref.float().abs().max().item():
                        print(f"   !! variant {v} differs from v1 on {name}: max|d| = {d:.4g}")
        ops.set_gemm_variant(0)
        ops.set_splitk(False)
        fl = 2.0 * M * N * K
        out[name] = {f"v{v}": dict(us=round(us, 1), tflops=round(fl / us / 1e6, 1)) for v, us in best.items()}
        print(name, M, N, K, out[name], flush=True)
    if "--small" in sys.argv or "--frames" in sys.argv:
        return
    # attention
    B, H, Nn, D = 16, 16, 577, 64
    qkv = rnd(B * Nn, 3 * H * D)
    o = torch.empty(B * Nn, H * D, dtype=torch.bfloat16, device=dev)
    st = (Nn * 3 * H * D, D, 3 * H * D)
    us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (Nn * H * D, D, H * D), B, H, Nn, Nn, 1, D ** -0.5, False, 0, D))
    out["attn_vit"] = dict(us=round(us, 1), tflops=round(4.0 * B * H * Nn * Nn * D / us / 1e6, 1))
    S, nh, nkv, D, smax = 1621, 32, 8, 128, 4096
    q, kc, vc = rnd(S, nh * D), rnd(nkv, smax, D), rnd(nkv, smax, D)
    o = torch.empty(S, nh * D, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: ops.attn_fwd(q, kc, vc, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D))
    out["attn_prefill"] = dict(us=round(us, 1), tflops=round(4.0 * nh * (S * (S + 1) / 2) * D / us / 1e6, 1))
    part = torch.empty(nh * 64 * 130, device=dev)
    od = torch.empty(nh * D, dtype=torch.bfloat16, device=dev)
    qkv1 = rnd((nh + 2 * nkv) * D)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.arange(smax).float()[:, None] * inv[None]
    cos_t, sin_t = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
    us = timeit(lambda: ops.attn_decode(qkv1, kc, vc, cos_t, sin_t, part, od, nh, nkv, 1650, D ** -0.5), iters=50)
    out["attn_decode_ctx1650"] = dict(us=round(us, 1), gbs=round(2 * 1650 * nkv * D * 2 / us / 1e3, 1))
    # decode gemv
    for name, N, K, kw in (("gemv_qkv", 6144, 4096, dict(norm=True)), ("gemv_wo", 4096, 4096, dict()), ("gemv_gateup", 28672, 4096, dict(swiglu=True, norm=True)),
                           ("gemv_down", 4096, 14336, dict()), ("gemv_lmhead", 32000, 4096, dict(norm=True, f32=True))):
        w, x = rnd(N, K, scale=K ** -0.5), rnd(K)
        nw = torch.randn(K, device=dev) if kw.get("norm") else None
        us = timeit(lambda: ops.gemv(w, x, norm_w=nw, swiglu=kw.get("swiglu", False), out_f32=kw.get("f32", False)), iters=50)
        out[name] = dict(us=round(us, 1), gbs=round(N * K * 2 / us / 1e3, 1))
    for k in ("attn_vit", "attn_prefill", "attn_decode_ctx1650", "gemv_qkv", "gemv_wo", "gemv_gateup", "gemv_down", "gemv_lmhead"):
        print(k, out[k], flush=True)
    print("JSON", json.dumps(out))


if __name__ == "__main__":
    main()
