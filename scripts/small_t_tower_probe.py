#!/usr/bin/env python
"""The CLIP tower on 1 / 2 / 4 / 8 / 16 frames with the row statistics reduced by their own launches (default) and inside the consuming GEMMs
(VL2_STAGE_SELF_REDUCE), alternating.  Usage: python scripts/small_t_tower_probe.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd import ops  # noqa: E402
from videollama2_amd.config import videollama2_7b  # noqa: E402
from videollama2_amd.tower import HipCLIPVisionTower  # noqa: E402
from videollama2_amd.weights import random_state_dict  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    cfg = videollama2_7b(16)
    sd = random_state_dict(cfg, dev, seed=1234, n_llm_layers=0)
    tower = HipCLIPVisionTower(cfg, sd, dev)
    side = cfg["vision"]["image_size"]
    frames = torch.randn((16, 3, side, side), generator=torch.Generator(device=dev).manual_seed(0), device=dev).bfloat16()
    for T in (1, 2, 4, 8, 16):
        res, outs = {0: [], 1: []}, {}
        for r in range(rounds + 1):
            for fl in (0, 1):
                ops.set_stage_flags(ops.STAGE_SELF_REDUCE if fl else 0)
                for _ in range(2):
                    outs[fl] = tower(frames[:T])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    tower(frames[:T])
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res[fl].append(e0.elapsed_time(e1) / 5)
        ops.set_stage_flags(0)
        print(f"tower T={T:2d}: finalize launches {min(res[0]):7.3f} ms   self-reduce in the consumers {min(res[1]):7.3f} ms   same bits {bool(torch.equal(outs[0], outs[1]))}", flush=True)


if __name__ == "__main__":
    main()
