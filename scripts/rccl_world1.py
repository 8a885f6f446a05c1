#!/usr/bin/env python
"""RCCL smoke on ONE GPU: a real `nccl` (= RCCL on ROCm) process group of world size 1, so that RCCL initialisation and every
collective the multi-GPU paths use (videollama2_amd/dist.py, decoder.py tensor parallelism) execute at least once on device
tensors on an MI355X, with the message shapes of BASELINE.json configs[2]/[3]:
    all_gather_into_tensor   [T/R, 576, 1024] bf16 visual tokens (north-star cut) and the sharded-connector cut's final tokens
    batch_isend_irecv        one-frame s1 halo [576, 4096] bf16 (to self at world 1)
    all_reduce               TP partial sums: prefill [1621, 4096] bf16, decode [4096] bf16 -- eager and captured in a hipGraph
Prints one JSON line; wall times are per call, averaged over `--iters` (at world 1 they measure launch + RCCL bookkeeping, not xGMI)."""
import argparse
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videollama2_amd.dist import FrameSharder  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / iters, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--port", type=int, default=29533)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"backend": "nccl (RCCL)", "world": 1}
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{args.port}", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=120))
    out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
    bf = dict(dtype=torch.bfloat16, device=dev)
    # ---- the sharder's collectives, through the product code (force the collective path at world 1)
    sh = FrameSharder()
    send = torch.randn((4, 576, 1024), **bf)
    recv = torch.empty((4, 576, 1024), **bf)
    out["all_gather_tokens_us"] = timed(lambda: sh._all_gather(recv, send), args.iters)
    out["all_gather_exact"] = bool(torch.equal(recv, send))
    halo_out = torch.randn((576, 4096), **bf)
    halo_in = torch.zeros((576, 4096), **bf)

    def p2p_self():
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, halo_out, 0), dist.P2POp(dist.irecv, halo_in, 0)])
        for r in reqs:
            r.wait()

    try:
        out["halo_isend_irecv_self_us"] = timed(p2p_self, args.iters)
        out["halo_exact"] = bool(torch.equal(halo_in, halo_out))
    except Exception as exc:          # a self send/recv is legal inside one RCCL group call; report rather than die if refused
        out["halo_isend_irecv_self_error"] = repr(exc)[:300]
    # ---- tensor-parallel all-reduces (decoder._reduce)
    big = torch.randn((1621, 4096), **bf)
    ref_big = big.clone()
    out["all_reduce_prefill_26MB_us"] = timed(lambda: dist.all_reduce(big), args.iters)
    out["all_reduce_identity_at_world1"] = bool(torch.equal(big, ref_big))
    small = torch.randn((4096,), **bf)
    out["all_reduce_decode_8KB_eager_us"] = timed(lambda: dist.all_reduce(small), args.iters)
    # ---- the same all-reduce captured in a hipGraph (what a graph-replayed TP decode step needs)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                dist.all_reduce(small)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(64):               # one decode token of a 32-layer TP decoder = 64 all-reduces
                dist.all_reduce(small)
        torch.cuda.synchronize()
        out["all_reduce_decode_8KB_graph_us"] = round(timed(g.replay, args.iters) / 64, 2)
        out["graph_capture_of_all_reduce"] = "ok"
    except Exception as exc:
        out["graph_capture_of_all_reduce"] = "failed: " + repr(exc)[:300]
    # ---- product paths on a small configuration (same code as the 7B model, narrow widths)
    try:
        out.update(product_checks(dev))
    except Exception as exc:
        out["product_checks_error"] = repr(exc)[:400]
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


def product_checks(dev):
    """(1) tensor-parallel decode step with its RCCL all-reduces CAPTURED in the decode hipGraph (group of one rank, reduces forced):
    tokens equal to the plain eager decoder.  (2) the rank-local encoder pieces replayed from their two hipGraphs (every rank's
    share run in this process): bit-identical to the eager launches, with timings of both."""
    from videollama2_amd.connector import HipSTCConnector
    from videollama2_amd.decoder import HipMistralDecoder
    from videollama2_amd.tower import HipCLIPVisionTower
    from videollama2_amd.weights import random_state_dict
    cfg = dict(vision=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4, image_size=112,
                           patch_size=14, layer_norm_eps=1e-5, select_layer=-2),
               llm=dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=2,
                        head_dim=128, vocab_size=4096, rms_norm_eps=1e-5, rope_theta=1e6), num_frames=8)
    cfg["llm"]["hidden_size"] = 1024          # q heads x head_dim = hidden (Mistral layout)
    sd = random_state_dict(cfg, dev, seed=7)
    res = {}
    x = torch.randn((40, cfg["llm"]["hidden_size"]), device=dev, dtype=torch.bfloat16, generator=torch.Generator(device=dev).manual_seed(1))
    plain = HipMistralDecoder(cfg, sd, dev, max_seq_len=128)
    want = plain.generate(x, max_new_tokens=12, use_graph=False)[0].tolist()
    tp = HipMistralDecoder(cfg, sd, dev, max_seq_len=128, tp_group=dist.group.WORLD)
    tp.tp_always_reduce = True
    got_eager = tp.generate(x, max_new_tokens=12, use_graph=False)[0].tolist()
    got_graph = tp.generate(x, max_new_tokens=12, use_graph=True)[0].tolist()
    res["tp_decode_graph_with_captured_all_reduce"] = "ok" if (tp.graph is not None and got_graph == want and got_eager == want) else \
        f"MISMATCH graph={got_graph} eager={got_eager} want={want}"
    tower = HipCLIPVisionTower(cfg, sd, dev)
    conn = HipSTCConnector(sd, dev)
    T = 8
    frames = torch.randn((T, 3, 112, 112), device=dev, dtype=torch.bfloat16, generator=torch.Generator(device=dev).manual_seed(2))
    for world in (2, 4):
        eager = FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world)
        sh = FrameSharder(use_graph=True)
        a = FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world, graphs=sh)
        b = FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world, graphs=sh)      # replay of the cached graphs
        t_e = timed(lambda: FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world), 5)
        t_g = timed(lambda: FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world, graphs=sh), 5)
        res[f"encoder_graphs_world{world}"] = dict(exact=bool(torch.equal(a, eager) and torch.equal(b, eager)), eager_us=t_e, graph_us=t_g)
    return res


if __name__ == "__main__":
    main()
