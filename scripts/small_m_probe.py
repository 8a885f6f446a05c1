#!/usr/bin/env python
"""The per-rank GEMM shapes of a frame-sharded encoder (2 frames per rank = 1154 rows; one image = 577 rows) on every tile form the product library
holds, forced through `variant`, interleaved (guide rule 24).  Usage: python scripts/small_m_probe.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402

SHAPES = [("vit q/k/v", 1154, 3072, 1024), ("vit out", 1154, 1024, 1024), ("vit fc1", 1154, 4096, 1024), ("vit fc2", 1154, 1024, 4096),
          ("stc s1 4096^2", 1152, 4096, 4096), ("stc s1 K=1024", 1152, 4096, 1024),
          ("vit q/k/v 1 img", 577, 3072, 1024), ("vit out 1 img", 577, 1024, 1024), ("vit fc1 1 img", 577, 4096, 1024), ("vit fc2 1 img", 577, 1024, 4096)]
VARIANTS = (0, 1, 32, 256, 4, 8, 12, 192, 224)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ops.attach_workspace("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        w = (torch.randn((N, K), device="cuda", generator=g) * K ** -0.5).bfloat16()
        out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        res, ref = {}, None
        for r in range(rounds + 1):
            for v in VARIANTS:
                try:
                    ops.set_gemm_variant(v)
                    for _ in range(3):
                        ops.gemm(a, w, out=out)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ops.gemm(a, w, out=out)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / 20)
                    elif v == 0:
                        ref = out.clone()
                    else:
                        assert torch.equal(out, ref), (name, v)
                except Exception as e:                                  # a tile form that is not built for the shape
                    res[v] = None
                finally:
                    ops.set_gemm_variant(0)
        fl = 2.0 * M * N * K
        print(f"{name:18s} {M}x{N}x{K}: " + "  ".join(f"v{v} {min(t):6.1f}us" if t else f"v{v}    n/a" for v, t in res.items()) +
              f"   | auto {fl / min(res[0]) / 1e6:5.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
