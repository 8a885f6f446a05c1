#!/usr/bin/env python
"""bench.py -- VideoLLaMA2-7B video-inference hot path on MI355X (HIP kernels via libvl2hip.so).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one synthetic 16-frame 336^2 video through the whole hot path with inputs resident in HBM:
  frames sharded over the N ranks: ViT + STC stage s1 per rank, one-frame halo, Conv3d/s2/readout on the rank's output
  frames, RCCL all-gather of the visual tokens (videollama2_amd/dist.py; N = 1 runs the same kernels unsharded)
  -> splice with 100 synthetic text ids -> Mistral-7B prefill (S = 1621) -> `--new-tokens` greedy decode steps
  (prefill/decode replicated on every rank: one sequence, no tensor parallelism in the reference either).
`value` = encoder video-frames/s (T / t_encode, t_encode = ViT + collectives + STC), BASELINE.json's headline; prefill and decode
throughput ride along as extra keys.  Weights: random-init bf16 of the exact VideoLLaMA2-7B architecture (no
checkpoints on the box); data: synthetic.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_FP8_TFLOPS = 5000.0    # dense fp8 (MX K = 128 measured 4647 TF), same table
PEAK_MFMA_BF16_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def algorithmic_tflop(cfg, T, S_text=100):
    """SURVEY.md 8(d) formulas (2*MAC; encoder layers beyond hidden_states[select_layer] are not counted), written over the
    config dict so the same accounting serves VideoLLaMA2-7B (CLIP / stc_connector / Mistral: 5.857 + 3.243 + 23.32 =
    32.42 TF at T=16) and VideoLLaMA2.1-7B (SigLIP / stc_connector_v35 / Qwen2).  Unpadded (real) shapes."""
    v, l = cfg["vision"], cfg["llm"]
    Dv, Iv, P = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    g = v["image_size"] // P
    npatch = g * g
    ntok = npatch + (0 if v.get("family", "clip") == "siglip" else 1)
    nrun = v["num_hidden_layers"] + 1 + v["select_layer"] if v["select_layer"] < 0 else v["select_layer"]
    vit = T * (2 * npatch * 3 * P * P * Dv + nrun * (ntok * 2 * (4 * Dv * Dv + 2 * Dv * Iv) + 4 * ntok * ntok * Dv))
    D, I, nl = l["hidden_size"], l["intermediate_size"], l["num_hidden_layers"]
    pad = 0 if cfg.get("projector", "stc_connector") == "stc_connector_v35" else 1
    o = lambda n: (n + 2 * pad - 2) // 2 + 1
    to, go = o(T), o(g)
    rd1, rd = int(round(Dv * 0.25)), int(round(D * 0.25))
    b1 = 2 * (npatch * (Dv * D * 2 + D * D) + npatch * D * 9 + 2 * D * rd1)
    b = 2 * (npatch * 2 * D * D + npatch * D * 9 + 2 * D * rd)
    nvis = to * go * go
    stc = T * (b1 + 3 * b) + 2 * nvis * D * D * 8 + to * 4 * 2 * (go * go * 2 * D * D + go * go * D * 9 + 2 * D * rd) + 2 * nvis * 2 * D * D
    S = nvis + S_text
    kvd = l["num_key_value_heads"] * l["head_dim"]
    qd = l["num_attention_heads"] * l["head_dim"]
    lin = nl * 2 * (D * qd * 2 + 2 * D * kvd + 3 * D * I)
    prefill = S * lin + nl * 4 * (S * (S + 1) // 2) * qd + 2 * D * l["vocab_size"]
    return vit / 1e12, stc / 1e12, prefill / 1e12, S


def decode_bytes_per_token(cfg, ctx):
    """bf16 weights streamed per token (embedding table excluded: one row) + KV read; SURVEY.md 8(d)."""
    l = cfg["llm"]
    D, I = l["hidden_size"], l["intermediate_size"]
    kvd, qd = l["num_key_value_heads"] * l["head_dim"], l["num_attention_heads"] * l["head_dim"]
    params = l["num_hidden_layers"] * (D * qd * 2 + 2 * D * kvd + 3 * D * I) + l["vocab_size"] * D
    return 2 * params + l["num_hidden_layers"] * 2 * kvd * 2 * ctx


def cpu_baseline(threads, T=16, S=1621):
    """The CPU restatement of the reference path (oracle/vl2_oracle.py, kind 'port': /root/reference does not exist on the GPU
    box) timed on this box's host cores in the three windows of SURVEY.md 8(d), each on a bounded slice of the SAME workload
    and scaled to it (the scaling is stated in `sample`):
        encode  : CLIP tower (23 layers) on 2 frames x T/2  +  STC connector on 4 frames x FLOP ratio to T frames
        prefill : 2 of the 32 decoder layers at the full S x 16  (+ final norm + lm_head on the last position, unscaled)
        decode  : 4 greedy steps through the same 2 layers x 16 (+ lm_head per step, unscaled)
    fp32 (the reference's CPU-runnable dtype).  The reference ITSELF was timed in the build container (8 cores,
    oracle/time_reference.py -> profiles/r02_cpu_reference.json); that record is attached as `reference_build_box`."""
    from oracle import vl2_oracle as O
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)

    def rnd(names_shapes, keep):
        sd = {}
        for name, shape in names_shapes:
            if not keep(name):
                continue
            fan = 1
            for d in shape[1:]:
                fan *= d
            t = torch.randn(shape, generator=g)
            sd[name] = t * (fan ** -0.5) if len(shape) >= 2 else (1.0 + 0.1 * t if name.endswith("weight") else 0.02 * t)
        return sd

    cfg = O.config_videollama2_7b(4)
    sd = rnd(O.state_dict_names(cfg), lambda n: "vision_tower" in n or "mm_projector" in n)
    fr = torch.randn(2, 3, 336, 336, generator=g)
    x = torch.randn(1, 4, 576, 1024, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.clip_tower(sd, cfg, fr)
        t1 = time.perf_counter()
        O.stc_connector(sd, x)
        t2 = time.perf_counter()
    _, stc4, _, _ = algorithmic_tflop(O.config_videollama2_7b(4), 4)
    _, stcT, _, _ = algorithmic_tflop(O.config_videollama2_7b(T), T)
    vit_s, stc_s = t1 - t0, t2 - t1
    enc_s = vit_s * (T / 2.0) + stc_s * (stcT / stc4)
    del sd
    cfg2 = O.config_videollama2_7b(T)
    nl_full, nl = cfg2["llm"]["num_hidden_layers"], 2
    cfg2["llm"]["num_hidden_layers"] = nl
    sd = rnd(O.state_dict_names(cfg2), lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head")))
    emb = 0.5 * torch.randn(S, cfg2["llm"]["hidden_size"], generator=g)
    n_dec = 4
    with torch.no_grad():
        t0 = time.perf_counter()
        logits, caches = O.mistral_forward(sd, cfg2, emb, 0, None)
        t1 = time.perf_counter()
        O.mistral_forward(sd, cfg2, emb[:1], 0, None, n_layers=0)            # final norm + lm_head alone (not scaled by depth)
        t_head = time.perf_counter() - t1
        t2 = time.perf_counter()
        pos = S
        for _ in range(n_dec):
            nxt = int(torch.argmax(logits[0]))
            xt = torch.nn.functional.embedding(torch.tensor([nxt]), sd["model.embed_tokens.weight"])
            logits, caches = O.mistral_forward(sd, cfg2, xt, pos, caches)
            pos += 1
        t3 = time.perf_counter()
    scale = nl_full / nl
    pre_s = max(t1 - t0 - t_head, 0.0) * scale + t_head
    dec_s = max((t3 - t2) / n_dec - t_head, 0.0) * scale + t_head
    out = dict(value=round(T / enc_s, 4), unit="frames/s", cores=threads, cores_present=os.cpu_count(), kind="port", where="GPU box host",
               sample=(f"oracle/vl2_oracle.py fp32, torch {threads} threads: encode = CLIP tower 2 frames ({vit_s:.2f} s) x {T // 2} + STC "
                       f"4 frames ({stc_s:.2f} s) x {stcT / stc4:.2f} (FLOP ratio to {T} frames); prefill = {nl} of {nl_full} decoder layers at S={S} x {scale:.0f} + lm_head; "
                       f"decode = {n_dec} greedy steps through {nl} layers x {scale:.0f} + lm_head"),
               encode=dict(s=round(enc_s, 2), frames_per_s=round(T / enc_s, 4)),
               prefill=dict(s=round(pre_s, 2), tokens_per_s=round(S / pre_s, 2)),
               decode=dict(s_per_token=round(dec_s, 3), tokens_per_s=round(1.0 / dec_s, 4)))
    rpath = os.path.join(ROOT, "profiles", "r02_cpu_reference.json")
    if os.path.exists(rpath):
        r = json.load(open(rpath))
        out["reference_build_box"] = dict(kind="reference", where="build container", source="profiles/r02_cpu_reference.json", cores=r.get("cores"),
                                          dtype=r.get("dtype"), encode_frames_per_s=r.get("encode_frames_per_s"),
                                          prefill_tokens_per_s=r.get("prefill_tokens_per_s"), decode_tokens_per_s=r.get("decode_tokens_per_s"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--model", choices=["v2", "v21", "72b"], default="v2",
                    help="v2 = VideoLLaMA2-7B (CLIP + stc_connector + Mistral-7B; BASELINE.json's metric), "
                         "v21 = VideoLLaMA2.1-7B-16F (SigLIP + stc_connector_v35 + Qwen2-7B; SURVEY 8f row 1), "
                         "72b = VideoLLaMA2-72B (CLIP + stc_connector + Qwen2-72B; 150 GB of weights on ONE GPU)")
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--llm-layers", type=int, default=None, help="debug only: fewer decoder layers (INVALID as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vit-only", action="store_true", help="skip the extra tower-only passes behind the `vit_only` key (profiling runs)")
    ap.add_argument("--u8-frames", action="store_true", help="(default since round 3; kept for old command lines) raw uint8 [T,S,S,3] frames")
    ap.add_argument("--bf16-frames", action="store_true",
                    help="feed randn bf16 [T,3,S,S] frames (rounds 1-2) instead of SURVEY 8d's recipe: np.random.default_rng(0) uint8 "
                         "[T,S,S,3] frames, resident in HBM, rescaled + normalised (the image processor's arithmetic tail, mm_utils.py:196-201) "
                         "in the patch-row kernel")
    ap.add_argument("--cut", choices=["north_star", "sharded_connector"], default="north_star",
                    help="with --gpus N > 1: the frame-parallel cut `value` is measured on (SURVEY 8e: north_star = BASELINE.json's own cut is "
                         "the default; the other cut and the ViT-only rate are reported beside it as extra keys)")
    ap.add_argument("--decode-batch", type=int, default=0,
                    help="extra measurement: after the timed steps, decode this many copies of the request TOGETHER (batched decode, "
                         "SURVEY 8f row 4) and report aggregate decode tokens/s as `batched_decode`")
    ap.add_argument("--prefill-batch", type=int, default=0,
                    help="extra measurement: encode + prefill this many copies of the request in ONE pass (throughput mode) and report "
                         "aggregate frames/s, prefill tokens/s and the MFMA fraction as `batched_prefill`")
    ap.add_argument("--tp", action="store_true", help="with --gpus N > 1: shard the LLM decoder tensor-parallel over the N ranks "
                                                      "(BASELINE.json configs[3]); default keeps the decoder replicated like the reference")
    ap.add_argument("--no-graph", action="store_true", help="eager decode loop instead of hipGraph replay")
    ap.add_argument("--no-encoder-graph", action="store_true",
                    help="with --gpus N > 1: launch the rank-local encoder pieces kernel by kernel instead of replaying their two hipGraphs")
    ap.add_argument("--vit-streams", type=int, default=None, help="ViT frames as N chunks on N HIP streams (default: the tower's own, 3)")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16", help="16-bit element type = which build of the library runs "
                    "(bf16: libvl2hip.so, BASELINE.json configs[1]; fp16: libvl2hip_f16.so, the reference's own mm_infer dtype)")
    ap.add_argument("--decode-weights", choices=["16bit", "fp8"], default="16bit",
                    help="fp8: ALSO time the decode steps on e4m3fn row-scaled copies of the decoder weights (SURVEY 8f row 5; an optional "
                         "arithmetic, reported under the `decode_fp8` key -- the headline decode numbers stay the 16-bit ones)")
    ap.add_argument("--prefill-weights", choices=["16bit", "fp8"], default="16bit",
                    help="fp8: ALSO time the prefill with its projections on the fp8 matrix pipe (W8A8: e4m3fn row-scaled weights, activations quantised "
                         "per token row; BASELINE.json configs[4] 'fp8 MFMA on CDNA4'); reported under the `prefill_fp8` key -- never the headline")
    ap.add_argument("--stage-flags", type=int, default=0, help="experiment controls of the stage-level calls (include/vl2hip.h VL2_STAGE_*: 1 persistent GEMM, "
                    "2 no mixed launch, 4 in-GEMM statistics reduction (ViT), 8 fused decode attention); travel in the call descriptors")
    ap.add_argument("--tune", type=str, default="", help="debug: comma list of gemm=<variant>, splitk=<0|1>, attn=<variant> (videollama2_amd/ops.py launch controls)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # VL2_DIST_BACKEND=gloo (debug): lets several ranks share one GPU, to exercise the multi-rank control flow on a 1-GPU box
    backend = os.environ.get("VL2_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)    # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from videollama2_amd import _lib, ops
    _lib.set_elem(args.dtype)                        # before any weight is packed: buffers and packed weights are allocated in it
    from videollama2_amd.config import videollama2_1_7b_16f, videollama2_72b, videollama2_7b
    from videollama2_amd.model import VideoLLaMA2Hip
    from videollama2_amd.weights import LazyRandomStateDict, random_state_dict

    if args.stage_flags:
        ops.set_stage_flags(args.stage_flags)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        {"gemm": ops.set_gemm_variant, "splitk": lambda x: ops.set_splitk(bool(x)), "attn": ops.set_attn_kv_groups}[k](int(v))
    T, n_new = args.frames, args.new_tokens
    cfg = {"v2": videollama2_7b, "v21": videollama2_1_7b_16f, "72b": videollama2_72b}[args.model](T)
    side = cfg["vision"]["image_size"]
    if args.model == "72b":      # generated parameter by parameter while packing: the raw + packed copies would not fit 288 GB
        sd = LazyRandomStateDict(cfg, dev, seed=1234, n_llm_layers=args.llm_layers)
    else:
        sd = random_state_dict(cfg, dev, seed=1234, n_llm_layers=args.llm_layers)
    tp_group = dist.group.WORLD if (args.tp and world > 1) else None
    model = VideoLLaMA2Hip(cfg, sd, dev, max_seq_len=4096, n_llm_layers=args.llm_layers, tp_group=tp_group)
    if tp_group is not None and backend != "nccl":
        args.no_graph = True                       # gloo debug mode: the all-reduces are staged through the host, not capturable
    del sd
    torch.cuda.empty_cache()
    model.sharder.use_graph = world > 1 and not args.no_encoder_graph     # rank-local ViT + s1 | conv3d + s2 + readout as two graphs
    if args.vit_streams is not None:
        model.vision_tower.streams = args.vit_streams

    model.sharder.cut = args.cut
    if args.bf16_frames:
        g = torch.Generator(device=dev).manual_seed(0)
        frames = torch.randn((T, 3, side, side), generator=g, device=dev, dtype=torch.float32).to(_lib.elem_dtype())
    else:        # SURVEY.md 8(d) synthetic input: seeded uint8 video frames (what process_video decodes / resizes to), uploaded ONCE
        import numpy as np
        frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (T, side, side, 3), dtype=np.uint8)).to(dev)
    V = cfg["llm"]["vocab_size"]
    cg = torch.Generator().manual_seed(1)
    ids = torch.cat([torch.tensor([1]), torch.randint(3, V, (31,), generator=cg), torch.tensor([-201]),
                     torch.randint(3, V, (68,), generator=cg)])[None].to(dev)
    mask = torch.ones_like(ids)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    graph = None

    def step(rec=None):
        e = [ev() for _ in range(4)]
        e[0].record()
        _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, mask, None, None, [(frames, "video")])
        e[1].record()
        dec = model.decoder
        dec.prefill(emb[0])                          # logits of the last position -> dec.logits
        e[2].record()
        if graph is not None:                        # one captured hipGraph per token: argmax + the whole decode step
            dec.state.copy_(torch.tensor([dec.pos - 1, 0], dtype=torch.int32), non_blocking=True)
            for s in range(n_new):                   # fixed length: stop criteria disabled for timing
                graph.replay()
            dec.pos += n_new
        else:
            for s in range(n_new):
                ops.argmax(dec.logits, dec.tok)
                dec.decode_step()
        e[3].record()
        if rec is not None:
            rec.append(e)
        return emb.shape[1]

    S = step()                                       # eager pass (also performs every first-launch initialisation)
    shard_check = None
    if world > 1:                                    # the sharded (+ graph-replayed) encoder must give the unsharded encoder's bits
        sharded = model.encode_images_or_videos([(frames, "video")])
        sharded2 = model.encode_images_or_videos([(frames, "video")])          # second call = replay of the cached graphs
        feats = model.vision_tower(frames)
        whole = model.mm_projector(feats.view(1, *feats.shape))
        ok = torch.tensor([int(torch.equal(sharded, whole) and torch.equal(sharded2, whole))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN) if backend == "nccl" else None
        shard_check = bool(ok.item())
        if not shard_check:
            raise SystemExit(f"rank {rank}: sharded encoder output differs from the unsharded encoder")
    if not args.no_graph:
        try:
            graph = model.decoder.capture_graph()
        except Exception:                            # a TP decode graph captures RCCL all-reduces: if the capture is refused on this
            if tp_group is None:                     # RCCL build, measure the eager loop instead of failing the run
                raise
            graph = None
    for _ in range(args.warmup):
        S = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rec = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        S = step(rec)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    ms_step = dt * 1e3 / args.steps
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in rec) / len(rec)
    pre_ms = sum(e[1].elapsed_time(e[2]) for e in rec) / len(rec)
    dec_ms = sum(e[2].elapsed_time(e[3]) for e in rec) / len(rec)

    # ---- the encoder alone: ViT-only frames/s (no collective, no connector) and, with N > 1 ranks, BOTH frame-parallel cuts
    #      (SURVEY 8e), each timed like the main loop (barrier + synchronize on both sides, max over ranks)
    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = tt.item()
        return d * 1e3 / iters

    enc_iters = max(args.steps, 3)
    parts = model.sharder.split(T, world)
    s0, c0 = parts[rank]
    # (--no-vit-only: a rocprofv3 trace of this command then holds whole steps only; the world == 1 line does not need the figure)
    vit_ms = None if args.no_vit_only else timed(lambda: model.vision_tower(frames[s0:s0 + c0]) if c0 > 0 else None, enc_iters)
    cuts = {}
    if world > 1:
        keep = model.sharder.cut
        for cut in model.sharder.CUTS:
            model.sharder.cut = cut
            try:                                   # the headline numbers above are already measured: an error in a comparison pass is reported, not fatal
                ms = timed(lambda: model.encode_images_or_videos([(frames, "video")]), enc_iters)
                cuts[cut] = dict(encode_ms=round(ms, 3), frames_per_s=round(T / (ms / 1e3), 1),
                                 taken=("sharded_connector" if cut == "sharded_connector" and model.sharder.can_shard_connector(T) else "north_star"))
            except Exception as exc:               # noqa: BLE001
                cuts[cut] = dict(error=f"{type(exc).__name__}: {exc}"[:300])
        model.sharder.cut = keep

    # ---- N > 1: what every rank saw and did, each on its OWN clock (the headline numbers are max-over-ranks): a first multi-GPU run explains itself
    per_rank = None
    if world > 1:
        def local_ms(fn, iters):
            fn()
            torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters
        me = {"rank": rank, "device": torch.cuda.get_device_name(dev), "device_index": dev_index, "frames": c0, "backend": backend}
        try:
            tower, conn = model.vision_tower, model.mm_projector
            me["vit_ms"] = round(local_ms(lambda: tower(frames[s0:s0 + c0]), enc_iters), 3) if c0 > 0 else 0.0
            if model.sharder.can_shard_connector(T):          # the rank-local piece of the sharded-connector cut: ViT + RegStage s1 on its frames
                me["vit_s1_ms"] = round(local_ms(lambda: model.sharder.local_s1_of(tower, conn, frames[s0:s0 + c0]), enc_iters), 3)
            maxc = max(pc for _, pc in parts)
            send = torch.zeros((maxc, tower.num_patches, tower.hidden_size), dtype=_lib.elem_dtype(), device=dev)
            recv = torch.empty((world * maxc, tower.num_patches, tower.hidden_size), dtype=_lib.elem_dtype(), device=dev)
            dist.barrier()
            me["gather_us"] = round(local_ms(lambda: model.sharder._all_gather(recv, send), 10) * 1e3, 1)     # the ONE collective of the north_star cut, alone
            me["gather_mb"] = round(recv.numel() * recv.element_size() / 1e6, 2)
        except Exception as exc:                               # noqa: BLE001  (diagnostics never fail the run)
            me["error"] = f"{type(exc).__name__}: {exc}"[:300]
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)

    # ---- optional: the same decode steps on fp8 (e4m3fn, one power-of-two scale per output row) copies of the decoder's weights
    decode_fp8 = None
    if args.decode_weights == "fp8" and world == 1 and tp_group is None:
        dec = model.decoder
        dec.enable_fp8_decode()
        g8 = None if args.no_graph else dec.capture_graph()
        best = None
        for it in range(args.steps + 1):
            _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, mask, None, None, [(frames, "video")])
            dec.prefill(emb[0])
            dec.state.copy_(torch.tensor([dec.pos - 1, 0], dtype=torch.int32), non_blocking=True)
            a, b = ev(), ev()
            a.record()
            for s in range(n_new):
                if g8 is not None:
                    g8.replay()
                else:
                    ops.argmax(dec.logits, dec.tok)
                    dec.decode_step()
            b.record()
            torch.cuda.synchronize()
            if g8 is not None:
                dec.pos += n_new
            if it:
                best = a.elapsed_time(b) if best is None else min(best, a.elapsed_time(b))
        l = cfg["llm"]
        kvd, qd = l["num_key_value_heads"] * l["head_dim"], l["num_attention_heads"] * l["head_dim"]
        rows = l["num_hidden_layers"] * (qd + 2 * kvd + l["hidden_size"] + 2 * l["intermediate_size"] + l["hidden_size"]) + l["vocab_size"]
        b16 = decode_bytes_per_token(cfg, S + n_new // 2)
        kv = l["num_hidden_layers"] * 2 * kvd * 2 * (S + n_new // 2)
        b8 = (b16 - kv) // 2 + rows * 4 + kv
        decode_fp8 = {"ms_per_token": round(best / n_new, 4), "tokens_per_s": round(n_new / (best / 1e3), 2),
                      "bytes_per_token": b8, "hbm_frac": round(b8 / (best / n_new / 1e3) / 1e9 / PEAK_HBM_GBS, 4),
                      "what": "decode projections + lm_head on OCP e4m3fn copies of the packed weights, one power-of-two fp32 scale per output row, "
                              "16-bit activations, fp32 accumulation (csrc/k_fp8.h; prefill keeps the 16-bit weights).  OPTIONAL arithmetic: not the "
                              "reference's, not the headline (`decode_ms_per_token` is the 16-bit path); tests/test_gpu_fp8.py, profiles/r04_fp8_parity.json"}
        dec.enable_fp8_decode(False)
        graph = None if args.no_graph else dec.capture_graph()      # the 16-bit graph again for the passes below

    # ---- optional: the prefill with its four projections per layer on the fp8 matrix pipe (v_mfma_f32_32x32x64_f8f6f4; W8A8)
    prefill_fp8 = None
    if args.prefill_weights == "fp8" and world == 1 and tp_group is None:
        dec = model.decoder
        _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, mask, None, None, [(frames, "video")])
        lref = dec.prefill(emb[0]).clone()
        dec.enable_fp8_prefill()
        l8 = dec.prefill(emb[0]).clone()
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = ev(), ev()
            e0.record()
            dec.prefill(emb[0])
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None or t < best else best
        ops.PROFILE = []
        dec.prefill(emb[0])
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        by = {}
        for pr in prof:
            if pr[0] == "gemm_fp8":
                e = by.setdefault(pr[4], [0, 0.0, 0.0])
                e[0] += 1; e[1] += pr[1] / 1e9; e[2] += pr[2].elapsed_time(pr[3])
        _, _, ptf, _ = algorithmic_tflop(cfg, T)
        prefill_fp8 = {"prefill_ms": round(best, 3), "prefill_tokens_per_s": round(S / (best / 1e3), 1),
                       "vs_16bit_prefill": round(pre_ms / best, 3),
                       "mfma_frac_of_fp8_peak": round(ptf / (best / 1e3) / PEAK_MFMA_FP8_TFLOPS, 4), "peak_tflops": PEAK_MFMA_FP8_TFLOPS,
                       "logits_rel_l2_vs_16bit_prefill": round(float((l8.float() - lref.float()).norm() / lref.float().norm()), 5),
                       "top1_equal": bool(int(l8.argmax()) == int(lref.argmax())),
                       "shapes": [dict(M=k[0], N=k[1], K=k[2], launches=v[0], avg_launch_us=round(1e3 * v[2] / v[0], 2), tflops=round(v[1] / v[2], 1))
                                  for k, v in sorted(by.items(), key=lambda kv: -kv[1][2])],
                       "arithmetic": "W8A8: OCP e4m3fn weights (one power-of-two scale per output row, the decode copies) x e4m3fn activations (one power-of-two "
                                     "scale per token row, quantised on the fly by vl2_quant_act_fp8, which also computes the RMS rstd), fp32 accumulation on "
                                     "v_mfma_f32_32x32x64_f8f6f4 (csrc/k_gemm.h gemm3 / gemm4 FP8); attention, RoPE, KV cache, lm_head 16-bit.  OPTIONAL "
                                     "arithmetic: not the reference's, not the headline (`prefill_ms` is the 16-bit path); oracle/fp8_oracle.py gemm_w8a8, "
                                     "tests/test_gpu_fp8.py"}
        dec.enable_fp8_prefill(False)
        dec.prefill(emb[0])

    # ---- roofline of the dominant kernel (gemm_bf16_kernel, MFMA-bound): one extra profiled pass, every GEMM launch
    #      bracketed by HIP events on the launch stream; achieved = sum(algorithmic FLOPs) / sum(kernel time)
    batched = None
    if args.decode_batch > 1 and world == 1:
        # the same spliced prompt, `decode_batch` times: prefill each copy, then time n_new batched decode steps
        _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, mask, None, None, [(frames, "video")])
        dec, nb = model.decoder, args.decode_batch
        bb = dec._ensure_batch(nb)
        for b in range(nb):
            dec.prefill(emb[0], cache=([k[b] for k in bb["k"]], [v[b] for v in bb["v"]]), logits_out=bb["logits"][b])
        bb["pos"][:nb].fill_(emb.shape[1])
        for b in range(nb):
            ops.argmax(bb["logits"][b], bb["tok"][b:b + 1])
        dec._decode_kernels_batched(nb)
        torch.cuda.synchronize()

        def bstep():
            for b in range(nb):
                ops.argmax(bb["logits"][b], bb["tok"][b:b + 1])
            dec._decode_kernels_batched(nb)

        bgraph = None
        if not args.no_graph:                      # positions / tokens live on the device: the whole batched step replays
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                bstep()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            bgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(bgraph, capture_error_mode="thread_local"):
                bstep()
            torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n_new):
            if bgraph is not None:
                bgraph.replay()
            else:
                bstep()
        e1.record()
        torch.cuda.synchronize()
        bms = e0.elapsed_time(e1) / n_new
        batched = {"sequences": nb, "launch": "eager" if bgraph is None else "hipGraph replay", "ms_per_step": round(bms, 4), "tokens_per_s": round(nb / (bms / 1e3), 1),
                   "hbm_frac_weights_once": round(decode_bytes_per_token(cfg, S + n_new // 2) / (bms / 1e3) / 1e9 / PEAK_HBM_GBS, 4)}

    bprefill = None
    if args.prefill_batch > 1 and world == 1:
        # throughput mode: `prefill_batch` copies of the video + prompt encoded in one tower call and prefilled in one pass
        nbp, dec = args.prefill_batch, model.decoder
        bb = dec._ensure_batch(nbp)
        caches = [([k[b] for k in bb["k"]], [v[b] for v in bb["v"]]) for b in range(nbp)]
        allf = torch.cat([frames] * nbp, 0)

        def bpass(e):
            e[0].record()
            tower = model.vision_tower(allf)
            t = tower.shape[0] // nbp
            embs = []
            for b in range(nbp):
                f = model.mm_projector(tower[b * t:(b + 1) * t].unsqueeze(0))
                _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, mask, None, None, [(frames, "video")], mm_features=f)
                embs.append(emb[0])
            e[1].record()
            dec.prefill_batch(embs, caches, bb["logits"][:nbp])
            e[2].record()

        bpass([ev(), ev(), ev()])
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e = [ev(), ev(), ev()]
            bpass(e)
            torch.cuda.synchronize()
            cur = (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]))
            best = cur if best is None or sum(cur) < sum(best) else best
        vtf, stf, ptf, _ = algorithmic_tflop(cfg, T)
        bprefill = {"videos": nbp, "encode_ms": round(best[0], 3), "prefill_ms": round(best[1], 3),
                    "frames_per_s": round(nbp * T / (best[0] / 1e3), 1), "prefill_tokens_per_s": round(nbp * S / (best[1] / 1e3), 1),
                    "forward_mfma_frac": round(nbp * (vtf + stf + ptf) / (sum(best) / 1e3) / PEAK_MFMA_BF16_TFLOPS, 4)}

    roof = None
    ops.PROFILE = [] if rank == 0 else None      # EVERY rank runs the extra pass (it contains the encoder's collectives);
    step()                                       # only rank 0 brackets its GEMM launches with events
    torch.cuda.synchronize()
    if rank == 0:
        prof = ops.PROFILE
        ops.PROFILE = None
        gflop = sum(p[1] for p in prof if p[0] == "gemm") / 1e9
        gms = sum(p[2].elapsed_time(p[3]) for p in prof if p[0] == "gemm")
        ngemm = sum(1 for p in prof if p[0] == "gemm")
        ach = gflop / gms if gms > 0 else 0.0                  # GFLOP/ms = TFLOP/s
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc run (scripts/gpu_traffic.sh; PMC cannot be
        # collected inside this process); only quoted for the workload it was collected on (T=16, 241 GEMM launches).
        traffic, tsrc = None, None
        for tname in ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json"):     # PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of this round's kernels first
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tname)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if T == 16 and args.model == "v2" and world == 1 and tj.get("launches_per_step") == ngemm:
                    traffic, tsrc = tj["hbm_bytes_per_launch"], "profiles/" + tname
                    break
        # the single dominant GEMM shape of the step (most total time): its own rate against the same peak
        by_shape = {}
        for pr in prof:
            if pr[0] == "gemm":
                e = by_shape.setdefault(pr[4], [0, 0.0, 0.0])
                e[0] += 1; e[1] += pr[1] / 1e9; e[2] += pr[2].elapsed_time(pr[3])
        dom = max(by_shape.items(), key=lambda kv: kv[1][2]) if by_shape else None
        kernels = ("every vl2_gemm call of the step, whichever kernel the library picks per shape: gemm_mix16_bf16_kernel (gate/up + SwiGLU since round 6: "
                   "256x256 ping-pong tiles + 128x128 tail tiles on v_mfma_f32_16x16x32 in ONE launch), gemm_mix_bf16_kernel (the same mixed launch on "
                   "v_mfma_f32_32x32x16: STC 4096-wide convs), gemm4_bf16_kernel (256x256 / 192x256 ping-pong), "
                   "gemm3_bf16_kernel (128x256 ping-pong), gemm7_bf16_kernel (fill-the-round 192x128 / 224x128: the M = 1521 connector shapes), "
                   "gemm6_bf16_kernel (persistent 256x256: the tower's q/k/v and fc1 since round 5), gemm_bf16_kernel (128x128), gemm_l8_bf16_kernel "
                   "(one-round 128x128), gemm_s_bf16_kernel (64x64)")
        roof = dict(bound="mfma", kernel=kernels, achieved=round(ach, 2), peak=PEAK_MFMA_BF16_TFLOPS,
                    unit="TFLOP/s", frac=round(ach / PEAK_MFMA_BF16_TFLOPS, 4), traffic=traffic,
                    launches=ngemm, avg_launch_us=round(1e3 * gms / max(ngemm, 1), 2),
                    flop_per_launch_avg=round(1e9 * gflop / max(ngemm, 1), 0), traffic_source=tsrc)
        # every GEMM shape of the step as timed IN the pipeline (HIP events around each launch on the launch stream)
        roof["shapes"] = [dict(M=k[0], N=k[1], K=k[2], launches=v[0], avg_launch_us=round(1e3 * v[2] / v[0], 2), tflops=round(v[1] / v[2], 1))
                          for k, v in sorted(by_shape.items(), key=lambda kv: -kv[1][2])]
        if dom is not None:
            (dM, dN, dK), (dn, dgf, dms) = dom
            roof["dominant"] = dict(gemm=f"M={dM} N={dN} K={dK}" + (f" (gate/up + SwiGLU: gemm_mix16_bf16_kernel = 256x256 ping-pong tiles (v_mfma_f32_16x16x32, 64-deep phases) on the first {dM // 256 * 256} rows + "
                                                                     f"128x128 tiles on the last {dM - dM // 256 * 256}, one launch)"
                                                                     if (dN, dK) == (2 * cfg["llm"]["intermediate_size"], cfg["llm"]["hidden_size"]) and dM % 256 else ""),
                                    launches=dn, avg_launch_us=round(1e3 * dms / dn, 2), gflop_per_launch=round(dgf / dn, 1),
                                    achieved=round(dgf / dms, 2), frac=round(dgf / dms / PEAK_MFMA_BF16_TFLOPS, 4), share_of_gemm_time=round(dms / gms, 3))

    if rank == 0:
        vit_tf, stc_tf, pre_tf, S_alg = algorithmic_tflop(cfg, T)
        fwd_ms = enc_ms + pre_ms
        if world == 1:
            par = "single GPU"
        else:
            if model.sharder.cut == "north_star" or not model.sharder.can_shard_connector(T):
                par = (f"frames sharded over {world} ranks, north_star cut: ViT per rank, ONE RCCL all-gather of [T/R,576,1024] visual tokens, "
                       f"connector replicated; ")
            else:
                par = (f"frames sharded over {world} ranks, sharded-connector cut (ViT + STC s1/conv3d/s2 per rank"
                       f"{' replayed from two hipGraphs' if model.sharder.use_graph else ''}, halo + RCCL all-gather of visual tokens); ")
            par += f"LLM {'tensor-parallel over the ranks' if args.tp else 'replicated'}"
        out = {
            "metric": {"v2": "video-frames/sec encoded (CLIP-ViT + STC), VideoLLaMA2-7B 16f@336^2; prefill/decode tokens/sec as extra keys",
                       "v21": "video-frames/sec encoded (SigLIP + STC v35), VideoLLaMA2.1-7B-16F 16f@384^2; prefill/decode tokens/sec as extra keys",
                       "72b": "video-frames/sec encoded (CLIP-ViT + STC-8192), VideoLLaMA2-72B 16f@336^2 on ONE GPU; prefill/decode tokens/sec as extra keys"}[args.model],
            "value": round(T / (enc_ms / 1e3), 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"VideoLLaMA2-7B, {T}-frame 336^2 video, {args.dtype}, S={S} prefill, {n_new} greedy decode tokens "
                                    f"(BASELINE.json configs[1])" if args.model == "v2" else
                                    f"VideoLLaMA2-72B (CLIP-ViT-L + stc_connector + Qwen2-72B, 74.9 B parameters resident on one MI355X), {T}-frame 336^2 video, "
                                    f"{args.dtype}, S={S} prefill, {n_new} greedy decode tokens (BASELINE.json configs[3] without the TP=8 split)" if args.model == "72b" else
                                    f"VideoLLaMA2.1-7B-16F (SigLIP-so400m-384 + stc_connector_v35 + Qwen2-7B), {T}-frame 384^2 video, {args.dtype}, "
                                    f"S={S} prefill, {n_new} greedy decode tokens (SURVEY 8f row 1; not BASELINE.json's metric config)"), "frames": T, "prefill_tokens": S, "new_tokens": n_new,
                       "parallelism": par,
                       "llm_layers": len(model.decoder.w["layers"]),
                       "decode": "eager launches" if graph is None else ("hipGraph replay (argmax + 32-layer step per token" + (", RCCL all-reduces captured)" if tp_group is not None else ")"))},
            "encode_ms": round(enc_ms, 3), "prefill_ms": round(pre_ms, 3), "decode_ms_per_token": round(dec_ms / n_new, 4),
            "prefill_tokens_per_s": round(S / (pre_ms / 1e3), 1), "decode_tokens_per_s": round(n_new / (dec_ms / 1e3), 2),
            "forward_tflop": round(vit_tf + stc_tf + pre_tf, 3),
            "forward_mfma_frac": round((vit_tf + stc_tf + pre_tf) / (fwd_ms / 1e3) / PEAK_MFMA_BF16_TFLOPS, 4),
            "decode_hbm_frac": round(decode_bytes_per_token(cfg, S + n_new // 2) / (dec_ms / n_new / 1e3) / 1e9 / PEAK_HBM_GBS, 4),
            "roofline": roof,
            "frames_input": "bf16 randn [T,3,S,S]" if args.bf16_frames else "uint8 [T,S,S,3] (np.random.default_rng(0)), normalised on the GPU in the patch-row kernel",
            "vit_only": None if vit_ms is None else {"ms": round(vit_ms, 3), "frames_per_s": round(T / (vit_ms / 1e3), 1),
                         "what": f"CLIP tower alone on the rank's {c0} of {T} frames (no collective, no connector), max over ranks"},
        }
        if world > 1:
            out["ranks_seen"] = sorted(r["rank"] for r in per_rank if r) if per_rank else None
            out["per_rank"] = per_rank
            out["targets"] = ("`value` (frames/s through ViT + collective + connector) answers BASELINE.json's metric; the >= 6x frame-parallel target of north_star can "
                              "only be met by the frame-parallel part -- `vit_only.frames_per_s` (no collective, no connector) -- because the connector is not frame-parallel "
                              "beyond the sharded_connector cut (SURVEY 8e); both cuts are timed below, `cut` names the one `value` used")
            out["cut"] = model.sharder.cut
            out["north_star_cut"] = cuts["north_star"]
            out["sharded_connector_cut"] = cuts["sharded_connector"]
        if shard_check is not None:
            out["sharded_encoder_equals_unsharded"] = shard_check
            out["encoder_graphs"] = "replayed" if model.sharder.use_graph else f"eager ({model.sharder.graph_error or 'disabled'})"
        if decode_fp8 is not None:
            out["decode_fp8"] = decode_fp8
        if prefill_fp8 is not None:
            out["prefill_fp8"] = prefill_fp8
        if batched is not None:
            out["batched_decode"] = batched
        if bprefill is not None:
            out["batched_prefill"] = bprefill
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32), T, S)   # eager torch oversubscribes badly beyond ~32 threads
            except Exception as exc:  # the oracle is a checker, never a dependency of the measured path
                out["cpu_baseline"] = {"value": None, "error": repr(exc)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
