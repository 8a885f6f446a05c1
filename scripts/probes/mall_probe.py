#!/usr/bin/env python
"""Probe (diagnostic): does a read by kernel A leave the bytes in the 256 MiB Infinity Cache for kernel B?  Times a streaming read
(the library's GEMV, weights [N, 4096] bf16) of a buffer of a given size cold (after flushing with 2 GiB of other reads) and warm
(immediately after another read of the same buffer).  Prints GB/s for both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videollama2_amd import ops
dev = "cuda"
flush = torch.ones(1 << 29, dtype=torch.int32, device=dev)
x = torch.randn(4096, device=dev).to(torch.bfloat16)
def t(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512):
    N = mb * (1 << 20) // (4096 * 2)
    w = (torch.randn(N, 4096, device=dev) * 0.02).to(torch.bfloat16)
    ops.gemv(w, x); torch.cuda.synchronize()
    cold, warm, warm_sum = [], [], []
    for r in range(5):
        flush.sum(); torch.cuda.synchronize()
        cold.append(t(lambda: ops.gemv(w, x)))
        warm.append(t(lambda: ops.gemv(w, x)))
        flush.sum(); torch.cuda.synchronize()
        w.view(torch.int32).sum(); torch.cuda.synchronize()     # a DIFFERENT kernel touches the bytes first
        warm_sum.append(t(lambda: ops.gemv(w, x)))
    by = N * 4096 * 2
    print(f"{mb:4d} MiB  cold {min(cold):7.1f} us = {by/min(cold)/1e6:6.2f} TB/s | warm (same kernel before) {min(warm):7.1f} us = {by/min(warm)/1e6:6.2f} TB/s | "
          f"warm (torch sum before) {min(warm_sum):7.1f} us = {by/min(warm_sum)/1e6:6.2f} TB/s", flush=True)
