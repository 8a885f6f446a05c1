"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (build container only: needs /root/reference).

Times the REAL reference (`Videollama2MistralForCausalLM`, imported in place through oracle/ref_harness.py with the four
shims of SURVEY.md 8c) on this container's host cores on the workload of BASELINE.json configs[1] -- VideoLLaMA2-7B
architecture, random-init weights, one synthetic 16-frame 336^2 video, S = 1621 prefill, greedy decode -- in the three
windows SURVEY.md 8(d) names:
    encode  = start of generate() .. first call of the decoder stack      (CLIP tower incl. its discarded 24th layer + STC + splice)
    prefill = first decoder-stack call .. second decoder-stack call       (32-layer prefill + lm_head + argmax + generate's bookkeeping)
    decode  = (second decoder-stack call .. end of generate()) / n        (n greedy steps)
all read off ONE `generate(max_new_tokens=1+n)` call through forward hooks on `model.get_model()` (MistralModel.forward): two
separately timed generate calls differ by more than a decode step on a shared host, hooks do not.
The result is committed as profiles/r02_cpu_reference.json and cited by bench.py's `cpu_baseline.reference_build_box`
(the GPU box has no /root/reference, so there bench.py times the port, oracle/vl2_oracle.py, instead).

    python -m oracle.time_reference [--frames 16] [--new-tokens 8] [--dtype float32|bfloat16] [--out profiles/r02_cpu_reference.json]
"""
import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--new-tokens", type=int, default=8)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference.json"))
    args = ap.parse_args()
    from oracle import ref_harness as RH
    from oracle import vl2_oracle as O
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    T = args.frames
    cfg = O.config_videollama2_7b(T)
    t0 = time.perf_counter()
    model, ref = RH.build_reference_model(cfg, seed=1234)
    dt = getattr(torch, args.dtype)
    if dt != torch.float32:
        model = model.to(dt)
    t_build = time.perf_counter() - t0
    frames = O.normalise_frames_u8(torch.randint(0, 256, (T, 336, 336, 3), dtype=torch.uint8,
                                                 generator=torch.Generator().manual_seed(0)).numpy()).to(dt)
    cg = torch.Generator().manual_seed(1)
    ids = torch.cat([torch.tensor([1]), torch.randint(3, 32000, (31,), generator=cg), torch.tensor([-201]),
                     torch.randint(3, 32000, (68,), generator=cg)])[None]
    mask = torch.ones_like(ids)
    n = args.new_tokens

    # ONE generate call; the three windows are read off timestamps taken by hooks on the decoder stack (MistralModel.forward):
    # its first call is the prefill, every later call one decode step; everything before the first call is the encoder
    # (encode_images_or_videos + the multimodal splice)
    stamps = []
    inner = model.get_model()
    h1 = inner.register_forward_pre_hook(lambda m, a: stamps.append(("start", time.perf_counter())))
    h2 = inner.register_forward_hook(lambda m, a, o: stamps.append(("end", time.perf_counter())))
    t_gen0 = time.perf_counter()
    with torch.inference_mode():
        out = model.generate(ids, attention_mask=mask, images=[(frames, "video")], do_sample=False, max_new_tokens=1 + n,
                             min_new_tokens=1 + n, use_cache=True, pad_token_id=0)
    t_gen1 = time.perf_counter()
    h1.remove(); h2.remove()
    starts = [t for k, t in stamps if k == "start"]
    ends = [t for k, t in stamps if k == "end"]
    n_calls = len(starts)
    t_enc = starts[0] - t_gen0
    t_pre = (starts[1] - starts[0]) if n_calls > 1 else (t_gen1 - starts[0])      # prefill forward + lm_head + argmax, up to the next step
    n_dec = n_calls - 1
    t_dec = ((t_gen1 - starts[1]) / n_dec) if n_dec > 0 else float("nan")
    with torch.inference_mode():
        feats = model.encode_images_or_videos([(frames[:2], "video")])            # shape law only (2 frames: cheap)
    n_vis = O.n_visual_tokens(T)
    S = n_vis + 100
    res = dict(
        what="the reference itself (/root/reference videollama2, HF transformers eager CPU kernels, timm restatement shim) on the build "
             "container's host cores; ONE generate() call, windows from hooks on the decoder stack's forward (first call = prefill, later calls = decode steps); no warm-up",
        workload=f"VideoLLaMA2-7B architecture, random-init {args.dtype}, {T} frames 336^2, S={S} prefill, {n} greedy decode tokens",
        cpu=platform.processor() or platform.machine(), cores=threads, torch_threads=torch.get_num_threads(), dtype=args.dtype,
        model_build_s=round(t_build, 1), encode_s=round(t_enc, 2), encode_frames_per_s=round(T / t_enc, 4),
        prefill_s=round(t_pre, 2), prefill_tokens_per_s=round(S / max(t_pre, 1e-9), 2),
        decode_s_per_token=round(t_dec, 3), decode_tokens_per_s=round(1.0 / max(t_dec, 1e-9), 4),
        generate_total_s=round(t_gen1 - t_gen0, 2), decoder_forward_calls=n_calls, new_tokens_returned=int(out.shape[1]), n_visual_tokens=n_vis,
        torch=torch.__version__)
    try:
        import transformers
        res["transformers"] = transformers.__version__
    except Exception:
        pass
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
