#!/usr/bin/env python
"""Timing of the STC connector's direct (non-GEMM) kernels at the T=16 shapes: depthwise 3x3 + LN + SiLU, SE squeeze / excite /
scale, the row LayerNorm with residual.  Usage: python scripts/stc_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

dev, C = "cuda", 4096
for name, Fr, H in (("s1 16x24x24", 16, 24), ("s2 9x13x13", 9, 13)):
    x = rnd(Fr * H * H, C)
    wt, lnw, lnb = torch.randn(9, C, device=dev) * 0.3, torch.randn(C, device=dev), torch.randn(C, device=dev)
    us = timeit(lambda: ops.dwconv3x3_ln_silu(x, wt, lnw, lnb, Fr, H, H))
    mb = 2 * Fr * H * H * C * 2 / 1e6
    print(f"dwconv_ln_silu {name}: {us:.1f} us, {mb / us * 1e-0:.2f} TB/s of x + y ({mb:.0f} MB)".replace("TB/s", "MB/us = TB/s"), flush=True)
    wtr = wt.to(x.dtype).float()                # taps as a checkpoint of the element type holds them
    us = timeit(lambda: ops.dwconv3x3_ln_silu_mean(x, wtr, lnw, lnb, Fr, H, H))
    print(f"dwconv_strip + squeeze (2 launches) {name}: {us:.1f} us, {mb / us:.2f} MB/us = TB/s of x + y", flush=True)
    us = timeit(lambda: ops.chan_mean(x, Fr, H * H))
    print(f"chan_mean {name}: {us:.1f} us", flush=True)
    m = ops.chan_mean(x, Fr, H * H)
    w1, b1 = rnd(C // 4, C, scale=C ** -0.5), torch.randn(C // 4, device=dev) * 0.1
    w2, b2 = rnd(C, C // 4, scale=1 / 32), torch.randn(C, device=dev) * 0.1
    us1 = timeit(lambda: ops.small_linear(m, w1, b1, ops.ACT_SILU))
    g1 = ops.small_linear(m, w1, b1, ops.ACT_SILU)
    us2 = timeit(lambda: ops.small_linear(g1, w2, b2, ops.ACT_SIGMOID))
    print(f"small_linear {name}: {us1:.1f} + {us2:.1f} us", flush=True)
    g2 = ops.small_linear(g1, w2, b2, ops.ACT_SIGMOID)
    us = timeit(lambda: ops.se_scale_(x, g2, Fr, H * H))
    print(f"se_scale {name}: {us:.1f} us", flush=True)
    us = timeit(lambda: ops.se_excite_scale_(x, g1, w2, b2, Fr, H * H))
    print(f"se_excite_scale (fc2 + sigmoid + scale, 1 launch) {name}: {us:.1f} us", flush=True)
    r = rnd(Fr * H * H, C)
    us = timeit(lambda: ops.layernorm(x, lnw, lnb, 1e-5, res=r, silu=True))
    print(f"layernorm+res+silu {name}: {us:.1f} us", flush=True)
