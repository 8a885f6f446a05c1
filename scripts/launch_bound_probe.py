#!/usr/bin/env python
"""Is the small-M (few frames per rank) encoder bound by kernel time or by host launch rate?  Enqueue time vs GPU time of
the tower + connector on 2 / 4 frames, eager and as a captured hipGraph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd.config import videollama2_7b
from videollama2_amd.connector import HipSTCConnector
from videollama2_amd.tower import HipCLIPVisionTower
from videollama2_amd.weights import random_state_dict
dev = torch.device("cuda", 0)
cfg = videollama2_7b(16)
sd = random_state_dict(cfg, dev, seed=1234, n_llm_layers=0)
tower, conn = HipCLIPVisionTower(cfg, sd, dev), HipSTCConnector(sd, dev)
del sd
for F in (2, 4, 16):
    frames = torch.randn((F, 3, 336, 336), device=dev).bfloat16()
    def run():
        f = tower(frames)
        return conn.run_s1(f.reshape(F * 576, -1), F, 24)
    for _ in range(3): run()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = run()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"frames {F}: eager enqueue {1e3*(t1-t0)/n:.3f} ms, eager total {1e3*(t2-t0)/n:.3f} ms, hipGraph replay {1e3*(t4-t3)/n:.3f} ms", flush=True)
