#!/usr/bin/env python
"""One GPU: the 16-frame encoder (tower + connector) launched kernel by kernel through the stage calls (what bench.py's N = 1 line times) against the same
launches replayed from ONE captured hipGraph -- are there enqueue gaps to take back?  Usage: python scripts/encoder_graph_probe.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd.config import videollama2_7b  # noqa: E402
from videollama2_amd.connector import HipSTCConnector  # noqa: E402
from videollama2_amd.tower import HipCLIPVisionTower  # noqa: E402
from videollama2_amd.weights import random_state_dict  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    cfg = videollama2_7b(16)
    sd = random_state_dict(cfg, dev, seed=1234, n_llm_layers=0)
    tower, conn = HipCLIPVisionTower(cfg, sd, dev), HipSTCConnector(sd, dev, padding=1)
    side = cfg["vision"]["image_size"]
    for T in (16, 8, 32):
        frames = torch.randn((T, 3, side, side), generator=torch.Generator(device=dev).manual_seed(0), device=dev).bfloat16()

        def enc():
            f = tower(frames)
            return conn(f.view(1, *f.shape))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                ref = enc()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = enc()
        except Exception as e:
            print(f"T={T}: capture failed: {e!r}"[:300])
            continue
        res = {"eager": [], "graph": []}
        for r in range(rounds + 1):
            for k in ("eager", "graph"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    if k == "eager":
                        enc()
                    else:
                        g.replay()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res[k].append(e0.elapsed_time(e1) / 5)
        print(f"encoder T={T}: eager {min(res['eager']):7.3f} ms   one hipGraph {min(res['graph']):7.3f} ms   same bits {bool(torch.equal(out, ref))}", flush=True)


if __name__ == "__main__":
    main()
