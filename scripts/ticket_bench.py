#!/usr/bin/env python
"""Producer-side finalize (k_gemm.h gemm_rows_ticket) against the separate row_norm_finalize launch, per statistics-producing GEMM of the T = 16
step, producer + consumer PAIR timed together (the launch sits between them): interleaved in one process on one box.
   launch:  GEMM(stats_out) -> row_norm_finalize -> consumer GEMM(row_norm)
   ticket:  GEMM(stats_out, row_norm_out)        -> consumer GEMM(row_norm)
Usage: python scripts/ticket_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from scripts.kernel_bench import rnd, timeit  # noqa: E402

dev = "cuda"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.attach_workspace(dev)
    for name, M, N, K, kind, NC, cact, sw in (("vit_out->fc1", 9232, 1024, 1024, 2, 4096, 1, False), ("vit_fc2->qkv", 9232, 1024, 4096, 2, 3072, 0, False),
                                              ("llm_o->gate/up", 1621, 4096, 4096, 1, 28672, 0, True), ("llm_down->qkv", 1621, 4096, 14336, 1, 6144, 0, False)):
        a, w, res = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(M, N)
        bias = torch.randn(N, device=dev) if kind == 2 else None
        wc = rnd(NC, N, scale=N ** -0.5)
        cbias = torch.randn(NC, device=dev) if kind == 2 else None
        cs = torch.randn(NC, device=dev) if kind == 2 else None
        st = torch.empty(M, N // 64, 2, device=dev)
        rn = torch.empty(M, 2, device=dev)
        tick = torch.zeros(M // 64 + 2, dtype=torch.int32, device=dev)
        x = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        y = torch.empty(M, NC // 2 if sw else NC, dtype=torch.bfloat16, device=dev)

        def launch():
            ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=x)
            ops.row_norm_finalize(st, N, kind, 1e-5, out=rn)
            ops.gemm(x, wc, bias=cbias, act=cact, swiglu=sw, out=y, norm=(kind, rn, 1e-5, cs))

        def ticket():
            ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=x, norm_out=(kind, 1e-5, rn, tick))
            ops.gemm(x, wc, bias=cbias, act=cact, swiglu=sw, out=y, norm=(kind, rn, 1e-5, cs))

        def producer_only():
            ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=x)

        def producer_ticket():
            ops.gemm(a, w, bias=bias, res=res, stats_out=st, out=x, norm_out=(kind, 1e-5, rn, tick))

        tab = {k: [] for k in ("launch", "ticket", "producer", "producer+ticket")}
        for _ in range(rounds):
            tab["launch"].append(timeit(launch, iters=30))
            tab["ticket"].append(timeit(ticket, iters=30))
            tab["producer"].append(timeit(producer_only, iters=30))
            tab["producer+ticket"].append(timeit(producer_ticket, iters=30))
        print(f"{name:16s} {M}x{N}x{K}: " + "  ".join(f"{k}: " + "/".join(f"{t:.1f}" for t in ts) + " us" for k, ts in tab.items()), flush=True)


if __name__ == "__main__":
    main()
