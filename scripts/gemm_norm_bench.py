#!/usr/bin/env python
"""What the norm-carrying GEMM chain costs and saves, per shape of the T=16 workload (interleaved A/B in one process):
   consumer:  plain GEMM on a pre-normalised input   vs   the same GEMM normalising in its epilogue from row statistics
   producer:  plain GEMM with the residual fused      vs   the same GEMM also emitting the row statistics
   and the standalone kernels the chain removes (vl2_layernorm / vl2_rmsnorm) or adds (vl2_row_stats)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

dev = "cuda"


def ab(fns, rounds=4, iters=20):
    best = {}
    for _ in range(rounds):
        for k, fn in fns.items():
            best[k] = min(best.get(k, 1e9), timeit(fn, iters=iters))
    return {k: round(v, 1) for k, v in best.items()}


def main():
    ops.attach_workspace(dev)
    out = {}
    for name, M, N, K, kind, kw in (("vit_qkv", 9232, 3072, 1024, 2, dict(bias=True)), ("vit_fc1", 9232, 4096, 1024, 2, dict(bias=True, act=1)),
                                    ("llm_qkv", 1621, 6144, 4096, 1, dict()), ("llm_gateup", 1621, 28672, 4096, 1, dict(swiglu=True))):
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        st = ops.row_stats(x)
        rn = ops.row_norm_finalize(st, K, kind, 1e-5)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        cs = torch.randn(N, device=dev) if kind == 2 else None
        c = torch.empty(M, N // 2 if kw.get("swiglu") else N, dtype=torch.bfloat16, device=dev)
        fns = {}
        for v in (0, 1, 4, 8):
            def plain(v=v):
                ops.set_gemm_variant(v)
                ops.gemm(x, w, bias=bias, act=kw.get("act", 0), swiglu=kw.get("swiglu", False), out=c)

            def fused(v=v):
                ops.set_gemm_variant(v)
                ops.gemm(x, w, bias=bias, act=kw.get("act", 0), swiglu=kw.get("swiglu", False), out=c, norm=(kind, st, 1e-5, cs))

            def final(v=v):
                ops.set_gemm_variant(v)
                ops.gemm(x, w, bias=bias, act=kw.get("act", 0), swiglu=kw.get("swiglu", False), out=c, norm=(kind, rn, 1e-5, cs))
            fns[f"plain_v{v}"], fns[f"fused_v{v}"], fns[f"final_v{v}"] = plain, fused, final
        out[name] = ab(fns)
        ops.set_gemm_variant(0)
        print(name, M, N, K, json.dumps(out[name]), flush=True)
    for name, M, N, K, kw in (("vit_wo", 9232, 1024, 1024, dict(bias=True)), ("vit_fc2", 9232, 1024, 4096, dict(bias=True)),
                              ("llm_wo", 1621, 4096, 4096, dict()), ("llm_down", 1621, 4096, 14336, dict())):
        a, w, res = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(M, N)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        st = torch.empty(M, N // 64, 2, device=dev)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out[name] = ab({"plain": lambda: ops.gemm(a, w, bias=bias, res=res, out=c),
                        "stats": lambda: ops.gemm(a, w, bias=bias, res=res, out=c, stats_out=st)})
        print(name, M, N, K, json.dumps(out[name]), flush=True)
    x1, x4 = rnd(9232, 1024), rnd(1621, 4096)
    w1, b1, w4 = torch.randn(1024, device=dev), torch.randn(1024, device=dev), torch.randn(4096, device=dev)
    s1, s4 = ops.row_stats(x1), ops.row_stats(x4)
    out["standalone"] = ab({"finalize_9232x16": lambda: ops.row_norm_finalize(s1, 1024, 2, 1e-5), "finalize_1621x64": lambda: ops.row_norm_finalize(s4, 4096, 1, 1e-5),
                            "layernorm_9232x1024": lambda: ops.layernorm(x1, w1, b1, 1e-5), "rmsnorm_1621x4096": lambda: ops.rmsnorm(x4, w4, 1e-5),
                            "row_stats_9232x1024": lambda: ops.row_stats(x1), "row_stats_1621x4096": lambda: ops.row_stats(x4)}, iters=50)
    print("standalone", json.dumps(out["standalone"]), flush=True)


if __name__ == "__main__":
    main()
