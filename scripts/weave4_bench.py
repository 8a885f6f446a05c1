#!/usr/bin/env python
"""WEAVE4 (k_gemm.h gemm4_body: the LDS-DMA of slab t+3 issued from the matrix phases) against the default load-phase issue, on the shapes of the T = 16
step that run on the 256 x 256 / 192 x 256 ping-pong bodies (gate/up and the STC 4096^2 convs through the mixed launch, ViT out_proj / fc2 and the
decoder's q/k/v on 192-row tiles) and on square reference shapes; the two forms INTERLEAVED in one process, warm operands; bit-identity asserted.
Usage: python scripts/weave4_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from videollama2_amd.weights import pack_gate_up  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

dev = "cuda"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.attach_workspace(dev)
    cases = []
    for name, M, N, K, kw in (("llm gate/up (mix)", 1621, 28672, 4096, dict(swiglu=True, rms=True)), ("llm q/k/v (192)", 1621, 6144, 4096, dict(rms=True)),
                              ("stc 4096^2 s1 (mix)", 9216, 4096, 4096, dict()), ("stc K=1024 (mix)", 9216, 4096, 1024, dict()),
                              ("vit fc2 (192)", 9232, 1024, 4096, dict(bias=True, res=True, stats=True)), ("vit out (192)", 9232, 1024, 1024, dict(bias=True, res=True, stats=True)),
                              ("vit q/k/v v8", 9232, 3072, 1024, dict(bias=True, v=8)), ("vit fc1 v8", 9232, 4096, 1024, dict(bias=True, act=1, v=8)),
                              ("sq 8192x4096x4096", 8192, 4096, 4096, dict()), ("sq 8192^3", 8192, 8192, 8192, dict())):
        a = rnd(M, K)
        w = pack_gate_up(rnd(N // 2, K, scale=K ** -0.5), rnd(N // 2, K, scale=K ** -0.5)) if kw.get("swiglu") else rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        ncol = N // 2 if kw.get("swiglu") else N
        res = rnd(M, ncol) if kw.get("res") else None
        so = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=dev) if kw.get("stats") else None
        norm = (ops.NORM_RMS, ops.row_norm_finalize(ops.row_stats(a), K, ops.NORM_RMS, 1e-5), 1e-5, None) if kw.get("rms") else None
        out = torch.empty((M, ncol), dtype=torch.bfloat16, device=dev)
        v = kw.get("v", 0)

        def fn(a=a, w=w, bias=bias, res=res, so=so, norm=norm, out=out, kw=kw, v=v):
            ops.set_gemm_variant(v)
            ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0), swiglu=bool(kw.get("swiglu")), stats_out=so, norm=norm, out=out)
            ops.set_gemm_variant(0)
        cases.append((name, M, N, K, fn, out))
    for name, M, N, K, fn, out in cases:
        base, wv = [], []
        ops.set_stage_flags(0)
        fn()
        ref = out.clone()
        ops.set_stage_flags(ops.STAGE_WEAVE4)
        fn()
        same = torch.equal(out, ref)
        for _ in range(rounds):
            ops.set_stage_flags(0)
            base.append(timeit(fn, iters=30))
            ops.set_stage_flags(ops.STAGE_WEAVE4)
            wv.append(timeit(fn, iters=30))
        ops.set_stage_flags(0)
        fl = 2.0 * M * N * K
        print(f"{name:22s} {M}x{N}x{K}: load-phase issue " + "/".join(f"{t:.1f}" for t in base) + f" us ({fl / min(base) / 1e6:.0f} TF/s)   woven " +
              "/".join(f"{t:.1f}" for t in wv) + f" us ({fl / min(wv) / 1e6:.0f} TF/s)   {min(wv) / min(base):.3f}x   bits equal: {same}", flush=True)


if __name__ == "__main__":
    main()
