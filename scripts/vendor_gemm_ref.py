#!/usr/bin/env python
"""Where the library's GEMM kernels stand against the vendor's: every GEMM shape of the T = 16 step, PLAIN (C = A W^T, bf16 in / out, no epilogue), this
library's kernel choice against torch.nn.functional.linear (= hipBLASLt / rocBLAS on ROCm), interleaved in one process on one box, warm operands.
A reference point only -- the product never calls the vendor GEMM (it has no fused epilogues, no gathered A, no norm carrying) -- but it says how much
of the distance to the MFMA peak is this library's and how much is the part's.  Usage: python scripts/vendor_gemm_ref.py [rounds]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

SHAPES = [("llm gate/up", 1621, 28672, 4096), ("llm down", 1621, 4096, 14336), ("llm q/k/v", 1621, 6144, 4096), ("llm o", 1621, 4096, 4096),
          ("vit fc1", 9232, 4096, 1024), ("vit fc2", 9232, 1024, 4096), ("vit q/k/v", 9232, 3072, 1024), ("vit out", 9232, 1024, 1024),
          ("stc 4096^2 (s1)", 9216, 4096, 4096), ("stc 4096^2 (s2)", 1521, 4096, 4096), ("stc K=1024", 9216, 4096, 1024), ("sq 8192x4096x4096", 8192, 4096, 4096),
          ("sq 8192^3", 8192, 8192, 8192)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.attach_workspace("cuda")
    tot = {"ours": 0.0, "vendor": 0.0}
    for name, M, N, K in SHAPES:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ours, vend, g8 = [], [], []
        for _ in range(rounds):
            ours.append(timeit(lambda: ops.gemm(a, w, out=out), iters=30))
            vend.append(timeit(lambda: F.linear(a, w), iters=30))
            # round 6: the same call on the 16 x 16 x 32 set (VL2_GEMM_MFMA16: gemm9 / the mixed launch / the one-round 128 x 128 body by the set's rule)
            g8.append(timeit(lambda: ops.gemm(a, w, out=out, mfma16=True), iters=30) if N % 256 == 0 else float("nan"))
        fl = 2.0 * M * N * K
        rel = (out.float() - F.linear(a, w).float()).norm() / out.float().norm()
        print(f"{name:20s} {M}x{N}x{K}: ours {min(ours):7.1f} us ({fl / min(ours) / 1e6:5.0f} TF/s)   vendor {min(vend):7.1f} us ({fl / min(vend) / 1e6:5.0f} TF/s)   "
              f"ours/vendor {min(ours) / min(vend):.2f}   16x16x32 set {min(g8):7.1f} us ({fl / min(g8) / 1e6:5.0f} TF/s)   rel diff {rel:.1e}", flush=True)
        if not name.startswith("sq"):
            tot["ours"] += min(ours); tot["vendor"] += min(vend)
    print(f"sum over the step's shapes (one launch each): ours {tot['ours']:.0f} us, vendor {tot['vendor']:.0f} us")


if __name__ == "__main__":
    main()
