#!/usr/bin/env python
"""gemm9 (csrc/k_gemm9.h: the 256 x 256 ping-pong tile on v_mfma_f32_16x16x32_bf16, variant 16) against the library's own choice and against the 256 x 256
kernel on v_mfma_f32_32x32x16_bf16 (variant 8), with the vendor's GEMM (torch.matmul -> hipBLASLt) beside them where the call is a plain one; the forms
INTERLEAVED in one process; rel-L2 against an fp32 matmul printed for each (variant 16 is not bit-identical with the family).
Usage: python scripts/mfma16_bench.py [rounds]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from videollama2_amd.weights import pack_gate_up  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

dev = "cuda"


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.attach_workspace(dev)
    for name, M, N, K, kw in (("llm gate/up", 1621, 28672, 4096, dict(swiglu=True, rms=True)), ("llm gate/up S=3242", 3242, 28672, 4096, dict(swiglu=True, rms=True)),
                              ("llm q/k/v", 1621, 6144, 4096, dict(rms=True)), ("llm down", 1621, 4096, 14336, dict(res=True)), ("llm o", 1621, 4096, 4096, dict(res=True)),
                              ("stc 4096^2", 9216, 4096, 4096, dict()), ("vit q/k/v", 9232, 3072, 1024, dict(bias=True)),
                              ("sq 8192x4096x4096", 8192, 4096, 4096, dict()), ("sq 8192^3", 8192, 8192, 8192, dict())):
        a = rnd(M, K)
        if kw.get("swiglu"):
            wg, wu = rnd(N // 2, K, scale=K ** -0.5), rnd(N // 2, K, scale=K ** -0.5)
            w = pack_gate_up(wg, wu)
        else:
            w = rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        ncol = N // 2 if kw.get("swiglu") else N
        res = rnd(M, ncol) if kw.get("res") else None
        rn = ops.row_norm_finalize(ops.row_stats(a), K, ops.NORM_RMS, 1e-5) if kw.get("rms") else None
        norm = (ops.NORM_RMS, rn, 1e-5, None) if rn is not None else None
        out = torch.empty((M, ncol), dtype=torch.bfloat16, device=dev)
        h = a.float() * rn[:, 1:2] if rn is not None else a.float()
        if kw.get("swiglu"):
            ref = F.silu(h @ wg.float().T) * (h @ wu.float().T)
        else:
            ref = F.linear(h, w.float(), bias) + (res.float() if res is not None else 0)

        def fn(v):
            ops.set_gemm_variant(v)
            ops.gemm(a, w, bias=bias, res=res, swiglu=bool(kw.get("swiglu")), norm=norm, out=out)
            ops.set_gemm_variant(0)
        forms = [("auto", lambda: fn(0)), ("v8", lambda: fn(8)), ("v16", lambda: fn(16))]
        plain = not kw.get("swiglu") and rn is None and res is None
        if plain:
            forms.append(("vendor", lambda: torch.matmul(a, w.T, out=out) if bias is None else torch.addmm(bias.bfloat16(), a, w.T, out=out)))
        errs = {}
        for nm, f in forms:
            f()
            errs[nm] = rel(out, ref)
        del ref, h
        ts = {nm: [] for nm, _ in forms}
        for _ in range(rounds):
            for nm, f in forms:
                ts[nm].append(timeit(f, iters=30))
        fl = 2.0 * M * N * K
        print(f"{name:20s} {M}x{N}x{K}: " + "   ".join(f"{nm} " + "/".join(f"{t:.1f}" for t in ts[nm]) + f" us ({fl / min(ts[nm]) / 1e6:.0f} TF/s, rel {errs[nm]:.2e})" for nm, _ in forms) +
              f"   v16/auto {min(ts['v16']) / min(ts['auto']):.3f}", flush=True)


if __name__ == "__main__":
    main()
