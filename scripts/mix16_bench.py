#!/usr/bin/env python
"""gate/up + SwiGLU (the step's dominant GEMM) on the two matrix instructions, interleaved on one box through the C ABI: the default's mixed launch
(256 x 256 ping-pong tiles on v_mfma_f32_32x32x16_bf16 + 128 x 128 tail tiles) against the round-6 twin on v_mfma_f32_16x16x32_bf16
(k_gemm9.h gemm_mix16_bf16_kernel, VL2_GEMM_MFMA16).  Usage: python scripts/mix16_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    K, N = 4096, 28672
    w = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).bfloat16()
    for M in (945, 1621, 2973):
        a = torch.randn((M, K), device=dev, generator=g).bfloat16()
        rn = ops.row_norm_finalize(ops.row_stats(a), K, ops.NORM_RMS, 1e-5)
        out = torch.empty((M, N // 2), device=dev, dtype=torch.bfloat16)
        res = {0: [], 16: [], 26: []}
        for r in range(rounds):
            for fl in (0, 16, 26):
                ops.set_gemm_variant(fl)
                for _ in range(3):
                    ops.gemm(a, w, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-5, None), out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.gemm(a, w, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-5, None), out=out)
                e1.record()
                torch.cuda.synchronize()
                res[fl].append(e0.elapsed_time(e1) * 1e3 / 20)
        ops.set_gemm_variant(0)
        fl = 2.0 * M * N * K
        for k, name in ((0, "32x32x16 mixed launch (family)"), (16, "16x16x32 mixed launch, 32-deep"), (26, "16x16x32 mixed launch, 64-deep")):
            t = sorted(res[k])[len(res[k]) // 2]
            print(f"gate/up M={M}: {name:34s} median {t:7.1f} us  {fl / t * 1e-6:7.1f} TF/s   all {[round(x, 1) for x in res[k]]}")


if __name__ == "__main__":
    main()
