#!/usr/bin/env python
"""Where the cycles of gemm9's two phases go (csrc/k_gemm9.h MODE 7 = variant 23: the default form with s_memtime stamps; sums per wave in the workspace).
Prints, per shape, the mean over all waves of cycles per slab for: LDS-DMA issue | fragment reads (issue ... returned, incl. the counted vmcnt wait) |
barrier behind the load phase | the 32 MFMAs' issue | barrier behind the matrix phase -- and the same split by wave group.  The matrix phase of one wave
is 32 x 16 = 512 matrix-pipe cycles; a slab costs a SIMD two of them.
Usage: python scripts/gemm9_phase_stamps.py OUT.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402

dev = "cuda:0"
NAMES = ("dma_issue", "reads_and_wait", "barrier_after_load", "mfma_issue", "barrier_after_mfma")


def main():
    out = {}
    ws = ops.attach_workspace(dev)
    for M, N, K in ((8192, 4096, 4096), (8192, 8192, 8192), (9232, 4096, 1024), (1621, 28672, 4096)):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        wgs = ((M + 255) // 256) * (N // 256)
        if wgs * 8 * 6 * 8 > ws.numel():
            print(f"{M}x{N}x{K}: workspace too small for {wgs} workgroups", flush=True)
            continue
        ops.set_gemm_variant(23)
        for _ in range(20):                      # warm: clocks settle at the power limit
            ops.gemm(a, w, out=c)
        torch.cuda.synchronize()
        ws.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, w, out=c)
        e1.record()
        torch.cuda.synchronize()
        ops.set_gemm_variant(0)
        t = ws[: wgs * 8 * 6 * 8].view(torch.int64).view(wgs, 8, 6).double().cpu()
        nt = t[..., 5]
        assert bool((nt == K // 32).all()), "stamps missing"
        per = t[..., :5] / nt[..., None]                                   # cycles per slab
        mean = per.mean((0, 1))
        g0, g1 = per[:, :4].mean((0, 1)), per[:, 4:].mean((0, 1))
        row = dict(us=e0.elapsed_time(e1) * 1e3, slab_cycles=float(mean.sum()), **{n: round(float(v), 1) for n, v in zip(NAMES, mean)},
                   group0={n: round(float(v), 1) for n, v in zip(NAMES, g0)}, group1={n: round(float(v), 1) for n, v in zip(NAMES, g1)})
        # variant 25: the undisturbed loop between ONE stamp pair -> cycles per slab without the stamps' own cost, and the EFFECTIVE shader clock:
        # (rounds of workgroups) x (cycles of a workgroup's K loop) against the wall time of a launch in a back-to-back loop
        ops.set_gemm_variant(25)
        for _ in range(200):
            ops.gemm(a, w, out=c)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            ops.gemm(a, w, out=c)
        e1.record()
        torch.cuda.synchronize()
        ops.set_gemm_variant(0)
        t2 = ws[: wgs * 8 * 6 * 8].view(torch.int64).view(wgs, 8, 6).double().cpu()
        loop_cyc = t2[..., 0].mean().item()
        setup_cyc, epi_cyc = t2[..., 1].mean().item(), t2[..., 2].mean().item()
        # the LAST launch's stamps: per CU slot the workgroups' [entry, end] intervals; s_memtime is one counter per XCD at best, so only spans are compared:
        # a workgroup's life = setup + loop + epilogue; kernel span on the device = max(end) - min(entry) over workgroups of an XCD is not comparable across XCDs
        life = (t2[..., 4] - t2[..., 3]).mean().item()
        us25 = e0.elapsed_time(e1) * 1e3 / 50
        rounds = wgs / 256.0
        row.update(loop_cycles_per_slab=round(loop_cyc / (K // 32), 1), us_per_call_loop=round(us25, 1), k_loop_cycles=round(loop_cyc), wg_rounds=rounds,
                   effective_clock_MHz_lower_bound=round(rounds * loop_cyc / us25, 1))
        row.update(setup_cycles=round(setup_cyc), epilogue_cycles=round(epi_cyc), workgroup_life_cycles=round(life),
                   effective_clock_MHz_from_lives=round(rounds * life / us25, 1))
        print(f"    a workgroup's life: setup {setup_cyc:.0f} + ring fill and K loop {loop_cyc:.0f} + epilogue {epi_cyc:.0f} = {life:.0f} cycles; {rounds:.2f} rounds x life / call time = "
              f"{row['effective_clock_MHz_from_lives']} MHz (what is missing to the sysfs clock = dispatch gaps and the tail of the last round)", flush=True)
        print(f"    undisturbed loop: {row['loop_cycles_per_slab']} cycles per slab (ideal 1024 = {1024 / row['loop_cycles_per_slab']:.1%} of the matrix pipe), {us25:.1f} us per call back to back, "
              f"{rounds:.2f} rounds x {loop_cyc:.0f} loop cycles / call time = {row['effective_clock_MHz_lower_bound']} MHz (lower bound of the effective clock: prologue / epilogue not counted)", flush=True)
        out[f"{M}x{N}x{K}"] = row
        print(f"{M}x{N}x{K}: {row['us']:.1f} us (stamped), cycles per slab and wave: " + "  ".join(f"{n} {row[n]}" for n in NAMES) + f"  = {row['slab_cycles']:.0f} (ideal 1024)", flush=True)
        print(f"    group 0: {row['group0']}\n    group 1: {row['group1']}", flush=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
