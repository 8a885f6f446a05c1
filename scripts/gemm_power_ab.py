#!/usr/bin/env python
"""Power and shader clock UNDER each GEMM form: is the vendor's kernel faster because it runs at a higher clock (less power per FLOP) or because it does more per
clock?  One process, one box: for each shape, each form (the 256 x 256 ping-pong kernel on v_mfma_f32_32x32x16_bf16 = variant 8; the same on
v_mfma_f32_16x16x32_bf16 = variant 16; the four-wave kernel = variant 9; the vendor's torch.matmul) is looped for `seconds`, alternating, while a thread samples
socket power and the shader clock from the amdgpu sysfs files at ~25 Hz.  Prints per form: us per call, TF/s, mean power, mean clock, and FLOP per clock per CU.
Usage: python scripts/gemm_power_ab.py OUT.json [--seconds 2.0] [--reps 2]"""
import argparse
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_trace import find_sysfs, sample_sysfs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--forms", default="", help="comma list of v8,v16,v17,...,v22,v26,v9,vendor (default: all)")
    args = ap.parse_args()
    from videollama2_amd import ops
    dev = "cuda"
    ops.attach_workspace(dev)
    card, hw = find_sysfs()
    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            s = sample_sysfs(card, hw) if card else {}
            s["t"] = time.time()
            samples.append(s)
            stop.wait(0.04)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    segs = []
    for M, N, K in ((8192, 4096, 4096), (8192, 8192, 8192), (9232, 4096, 1024)):
        a = (torch.randn(M, K, device=dev)).bfloat16()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def lib(v):
            def f():
                ops.set_gemm_variant(v)
                ops.gemm(a, w, out=out)
                ops.set_gemm_variant(0)
            return f
        forms = [("v8 32x32x16", lib(8)), ("v16 16x16x32", lib(16)), ("v17 16x16x32 dma-behind-reads", lib(17)), ("v18 16x16x32 woven", lib(18)), ("v19 16x16x32 register-staged", lib(19)), ("v20 16x16x32 dma 2+2", lib(20)), ("v21 16x16x32 dma 1+3", lib(21)), ("v22 16x16x32 dma 3+1", lib(22)), ("v26 16x16x32 64-deep phases", lib(26)), ("v9 4-wave", lib(9)),
                 ("vendor", lambda: torch.matmul(a, w.T, out=out))]
        forms = [f for f in forms if not args.forms or f[0].split()[0] in args.forms.split(",")]
        lib(16)()
        ref16 = out.clone()
        for v in (17, 18, 19, 20, 21, 22, 26):
            lib(v)()
            print(f"{M}x{N}x{K}: variant {v} bits equal to variant 16: {torch.equal(out, ref16)}", flush=True)
        del ref16
        for rep in range(args.reps):
            for name, fn in forms:
                fn()
                torch.cuda.synchronize()
                t0, n = time.time(), 0
                while time.time() - t0 < args.seconds:
                    for _ in range(50):
                        fn()
                    n += 50
                    torch.cuda.synchronize()
                t1 = time.time()
                segs.append(dict(shape=[M, N, K], form=name, rep=rep, t0=t0, t1=t1, calls=n, us_per_call=(t1 - t0) * 1e6 / n))
    stop.set()
    th.join()
    for sg in segs:
        inside = [s for s in samples if sg["t0"] + 0.4 <= s["t"] <= sg["t1"] - 0.1]
        pw = [s.get("power_uW", s.get("power_in_uW")) for s in inside if s.get("power_uW") or s.get("power_in_uW")]
        ck = [s["sclk_hz"] for s in inside if s.get("sclk_hz")]
        M, N, K = sg["shape"]
        tf = 2.0 * M * N * K / sg["us_per_call"] / 1e6
        clk = sum(ck) / len(ck) / 1e6 if ck else None
        sg.update(samples=len(inside), tflops=round(tf, 1), power_W_mean=round(sum(pw) / len(pw) / 1e6, 1) if pw else None, sclk_MHz_mean=round(clk, 1) if clk else None,
                  flop_per_clk_per_cu=round(tf * 1e12 / (clk * 1e6) / 256, 1) if clk else None)
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in sg.items() if k not in ("t0", "t1")}), flush=True)
    json.dump(dict(what=__doc__.split("\n")[0], source="sysfs" if card else "none", segments=segs), open(args.out, "w"))


if __name__ == "__main__":
    main()
