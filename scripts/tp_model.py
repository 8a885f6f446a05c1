#!/usr/bin/env python
"""Tensor-parallel decoder on a ONE-GPU box: rank 0's shard of the Mistral-7B / Qwen2-7B / Qwen2-72B decoder at TP = 1 / 2 / 4 / 8 run
alone (no collectives: `tp_shard`), prefill (S = 1621) and decode steps timed with HIP events.  The all-reduces are modelled
separately from message size and the xGMI link rate (full mesh, 7 links x ~153 GB/s per GPU: a direct reduce-scatter +
all-gather moves 2 x bytes/N over each link) plus a fixed launch/sync cost per collective.

    python scripts/tp_model.py [--model v2|v21] [--tps 1 2 4 8]
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videollama2_amd.config import videollama2_1_7b_16f, videollama2_72b, videollama2_7b
from videollama2_amd.decoder import HipMistralDecoder
from videollama2_amd.weights import LazyRandomStateDict, random_state_dict

LINK_GBS, COLL_FIXED_US = 153.0, 20.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["v2", "v21", "72b"], default="v2")
    ap.add_argument("--tps", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--tokens", type=int, default=1621)
    ap.add_argument("--new", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = {"v2": videollama2_7b, "v21": videollama2_1_7b_16f, "72b": videollama2_72b}[args.model](16)
    l = cfg["llm"]
    keep = ("model.layers.", "model.norm", "model.embed_tokens", "lm_head")
    if args.model == "72b":
        sd = LazyRandomStateDict(cfg, dev, seed=1234)      # every shard is generated while it is packed
    else:
        sd = {k: v for k, v in random_state_dict(cfg, dev, seed=1234).items() if k.startswith(keep)}
    x = (torch.randn(args.tokens, l["hidden_size"], device=dev) * 0.5).bfloat16()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    base = None
    for tp in args.tps:
        if l["num_key_value_heads"] % tp:
            continue
        dec = HipMistralDecoder(cfg, sd, dev, max_seq_len=2048, tp_shard=(0, tp))
        best_p, best_d = 1e9, 1e9
        for it in range(4):
            a, b, c = ev(), ev(), ev()
            a.record(); dec.prefill(x); b.record()
            for _ in range(args.new):
                dec.decode_step()
            c.record(); torch.cuda.synchronize()
            if it:
                best_p, best_d = min(best_p, a.elapsed_time(b)), min(best_d, b.elapsed_time(c) / args.new)
        nl, D = l["num_hidden_layers"], l["hidden_size"]
        ar_pre = 0.0 if tp == 1 else COLL_FIXED_US + 2 * (args.tokens * D * 2 / tp) / (LINK_GBS * 1e3)
        ar_dec = 0.0 if tp == 1 else COLL_FIXED_US + 2 * (D * 2 / tp) / (LINK_GBS * 1e3)
        pre, decms = best_p + 2 * nl * ar_pre / 1e3, best_d + 2 * nl * ar_dec / 1e3
        base = base or (pre, decms)
        row = dict(model=args.model, tp=tp, prefill_tokens=args.tokens, prefill_compute_ms=round(best_p, 3), prefill_allreduce_ms=round(2 * nl * ar_pre / 1e3, 3),
                   prefill_ms=round(pre, 3), prefill_speedup=round(base[0] / pre, 2), decode_compute_ms=round(best_d, 4),
                   decode_allreduce_ms=round(2 * nl * ar_dec / 1e3, 4), decode_ms_per_token=round(decms, 4), decode_speedup=round(base[1] / decms, 2))
        print(json.dumps(row), flush=True)
        del dec
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
