"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import torch


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# tolerances, stated once (rel-L2 against the fp32 oracle on identical bf16-rounded inputs/weights):
TOL_F32_OUT = 1e-3     # fp32-accumulate kernels with fp32 output
TOL_BF16_OUT = 3e-3    # same + one bf16 output rounding (floor 2^-9/sqrt(3) = 1.1e-3) and bf16 P in attention


def token_tie_ok(ours_logits, ref_logits):
    """The ONLY excuse for a greedy token that differs from the reference's: the reference's own top-2 margin at that step is
    smaller than twice the largest logit error of the path under test (a tie the bf16 floor cannot resolve).  Returns
    (ok, margin, dmax)."""
    a, b = ours_logits.detach().float().cpu(), ref_logits.detach().float().cpu()
    top2 = b.topk(2).values
    margin = (top2[0] - top2[1]).item()
    dmax = (a - b).abs().max().item()
    return margin < 2.0 * dmax, margin, dmax


def bf16_round(t):
    return t.bfloat16().float()


def sd_to(sd, dtype):
    return {k: (v.to(dtype) if torch.is_floating_point(v) else v) for k, v in sd.items()}


class ToyTokenizer:
    """Whitespace tokenizer with the attributes mm_infer / KeywordsStoppingCriteria touch (no tokenizer files on the box)."""
    eos_token, eos_token_id, bos_token_id, unk_token, pad_token, pad_token_id = "</s>", 2, 1, "<unk>", None, 0

    def __init__(self, vocab):
        self.vocab, self.prompts = vocab, []

    def _id(self, w):
        return 3 + (sum(ord(c) * (i + 7) for i, c in enumerate(w)) % (self.vocab - 3))

    def __call__(self, text, add_special_tokens=True):
        ids = [2 if w == "</s>" else self._id(w) for w in text.split()]
        return type("Enc", (), {"input_ids": ([1] if add_special_tokens else []) + ids})()

    def apply_chat_template(self, message, tokenize=False, add_generation_prompt=True):
        self.prompts.append(message)
        return "".join(f"[{m['role']}] {m['content']} " for m in message) + "[assistant]"

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(f"t{int(t)}" for t in row if not (skip_special_tokens and int(t) in (0, 1, 2))) for row in ids]


def write_synthetic_checkpoint(path, cfg, seed, oracle):
    """A local VideoLLaMA2 checkpoint directory (config.json + vision_config.json + model.safetensors, bf16, seeded weights of
    oracle.seeded_state_dict) as `videollama2_amd.api.model_init` / the reference's `model_init` read it.  Returns the HF dict."""
    import json
    import os
    from safetensors.torch import save_file
    v, l = cfg["vision"], cfg["llm"]
    sd = {k: t.bfloat16().contiguous() for k, t in oracle.seeded_state_dict(cfg, seed).items()}
    save_file(sd, os.path.join(str(path), "model.safetensors"))
    hf = dict(model_type="videollama2_qwen2" if oracle.llm_family(cfg) == "qwen2" else "videollama2_mistral",
              hidden_size=l["hidden_size"], intermediate_size=l["intermediate_size"], num_hidden_layers=l["num_hidden_layers"],
              num_attention_heads=l["num_attention_heads"], num_key_value_heads=l["num_key_value_heads"], head_dim=l["head_dim"],
              vocab_size=l["vocab_size"], rms_norm_eps=l["rms_norm_eps"], rope_theta=l["rope_theta"], num_frames=cfg.get("num_frames", 4),
              mm_vision_tower="somewhere/" + ("siglip-synthetic" if oracle.vision_family(cfg) == "siglip" else "clip-synthetic"),
              mm_projector_type=cfg.get("projector", "stc_connector"), mm_vision_select_layer=v["select_layer"])
    json.dump(hf, open(os.path.join(str(path), "config.json"), "w"))
    json.dump({k: v[k] for k in v if k != "select_layer"}, open(os.path.join(str(path), "vision_config.json"), "w"))
    return hf
