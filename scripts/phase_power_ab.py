#!/usr/bin/env python
"""Phase-resolved power / clock trace: is the chip at its power limit during the ViT, and what does the persistent GEMM do
to the clock?  ONE process on one box: the 16-frame 7B model of bench.py; for each phase (CLIP tower | STC connector | decoder prefill) and each
stage-flag setting, alternating, the phase alone is looped for `--seconds` while a thread samples socket power, shader clock and gpu_busy of GPU 0
from the amdgpu sysfs files at ~25 Hz.  A sample belongs to the segment it falls into (segments are seconds long, samples 40 ms apart: the phases are
resolved, which the 40-ms step of bench.py is not).  Output: one JSON with per-segment ms per pass, mean / max power, mean / min clock.
Usage: python scripts/phase_power_ab.py OUT.json [--flags 0,1] [--seconds 2.5] [--reps 2]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from power_trace import find_sysfs, sample_sysfs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--flags", default="0,1")
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--phases", default="vit,stc,prefill")
    args = ap.parse_args()
    flags = [int(f) for f in args.flags.split(",")]
    from videollama2_amd import ops
    from videollama2_amd.config import videollama2_7b
    from videollama2_amd.model import VideoLLaMA2Hip
    from videollama2_amd.weights import random_state_dict
    dev = torch.device("cuda", 0)
    T = args.frames
    cfg = videollama2_7b(T)
    sd = random_state_dict(cfg, dev, seed=1234)
    model = VideoLLaMA2Hip(cfg, sd, dev, max_seq_len=4096)
    del sd
    torch.cuda.empty_cache()
    side = cfg["vision"]["image_size"]
    frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (T, side, side, 3), dtype=np.uint8)).to(dev)
    feats = model.vision_tower(frames)
    feats = feats.view(1, *feats.shape)
    vis = model.mm_projector(feats)
    S = vis.shape[1] + 100
    emb = (torch.randn((S, cfg["llm"]["hidden_size"]), device=dev) * 0.02).to(feats.dtype)
    phases = {"vit": lambda: model.vision_tower(frames), "stc": lambda: model.mm_projector(feats), "prefill": lambda: model.decoder.prefill(emb)}
    phases = {k: v for k, v in phases.items() if k in args.phases.split(",")}
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
    for name, fn in phases.items():
        for f in flags:
            ops.set_stage_flags(f)
            fn()
        torch.cuda.synchronize()
        for rep in range(args.reps):
            for f in flags:
                ops.set_stage_flags(f)
                fn()
                torch.cuda.synchronize()
                t0, n = time.time(), 0
                while time.time() - t0 < args.seconds:
                    for _ in range(10):
                        fn()
                    n += 10
                    torch.cuda.synchronize()
                t1 = time.time()
                segs.append(dict(phase=name, stage_flags=f, rep=rep, t0=t0, t1=t1, passes=n, ms_per_pass=(t1 - t0) * 1e3 / n))
    ops.set_stage_flags(0)
    stop.set()
    th.join()
    for sg in segs:
        inside = [s for s in samples if sg["t0"] + 0.3 <= s["t"] <= sg["t1"] - 0.1]        # the first 0.3 s: the clock is still settling
        pw = [s.get("power_uW", s.get("power_in_uW")) for s in inside if s.get("power_uW") or s.get("power_in_uW")]
        ck = [s["sclk_hz"] for s in inside if s.get("sclk_hz")]
        sg.update(samples=len(inside), power_W_mean=round(sum(pw) / len(pw) / 1e6, 1) if pw else None, power_W_max=round(max(pw) / 1e6, 1) if pw else None,
                  sclk_MHz_mean=round(sum(ck) / len(ck) / 1e6, 1) if ck else None, sclk_MHz_min=round(min(ck) / 1e6, 1) if ck else None)
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in sg.items() if k not in ("t0", "t1")}))
    cap = samples[0].get("cap_uW") if samples else None
    json.dump(dict(what=__doc__.split("\n")[0], source="sysfs" if card else "none", cap_W=cap / 1e6 if cap else None, frames=T, segments=segs,
                   samples=[[round(s["t"] - samples[0]["t"], 3), round(s.get("power_uW", s.get("power_in_uW", 0)) / 1e6, 1), round(s.get("sclk_hz", 0) / 1e6), s.get("busy")]
                            for s in samples]), open(args.out, "w"))


if __name__ == "__main__":
    main()
