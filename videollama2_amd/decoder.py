"""HipMistralDecoder -- the decoder backend behind `Videollama2MistralForCausalLM.generate(inputs, images=...)`
(videollama2/model/videollama2_mistral.py:110-144): consumes `inputs_embeds [S, hidden]`, runs the Mistral prefill and
the greedy decode loop on its own KV cache and returns the NEW token ids, as HF `generate` does on the
`inputs_embeds` path.  Per-layer math follows HF:models/mistral/modeling_mistral.py (RMSNorm fp32 statistics,
rotate-half RoPE theta=1e6, causal GQA attention, SwiGLU), executed by libvl2hip.so kernels only."""
import torch
import torch.nn as nn

from . import ops
from .weights import pack_decoder


class HipMistralDecoder(nn.Module):
    def __init__(self, cfg, state_dict, device="cuda", max_seq_len=4096, n_layers=None):
        super().__init__()
        self.cfg = cfg
        l = cfg["llm"]
        self._dev = torch.device(device)
        self.w = pack_decoder(state_dict, cfg, self._dev, n_layers)
        self.n_layers = len(self.w["layers"])
        self.nh, self.nkv, self.hd = l["num_attention_heads"], l["num_key_value_heads"], l["head_dim"]
        self.D, self.V, self.eps = l["hidden_size"], l["vocab_size"], l["rms_norm_eps"]
        self.max_seq_len = max_seq_len
        # RoPE tables, fp32, HF MistralRotaryEmbedding: inv_freq = theta^(-2i/d); cos/sin of pos*inv_freq
        inv = 1.0 / (l["rope_theta"] ** (torch.arange(0, self.hd, 2, dtype=torch.int64).float() / self.hd))
        fr = torch.arange(max_seq_len, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos_t = fr.cos().contiguous().to(self._dev)
        self.sin_t = fr.sin().contiguous().to(self._dev)
        bf = dict(dtype=torch.bfloat16, device=self._dev)
        self.kcache = [torch.zeros((self.nkv, max_seq_len, self.hd), **bf) for _ in range(self.n_layers)]
        self.vcache = [torch.zeros((self.nkv, max_seq_len, self.hd), **bf) for _ in range(self.n_layers)]
        self.decode_chunk = 64
        nsplit_max = (max_seq_len + self.decode_chunk - 1) // self.decode_chunk
        self.partial = torch.empty((self.nh * nsplit_max * 130,), dtype=torch.float32, device=self._dev)
        self.tok = torch.zeros((1,), dtype=torch.int32, device=self._dev)
        self.pos = 0

    # ------------------------------------------------------------------ prefill (M = S tokens, MFMA GEMMs)
    @torch.no_grad()
    def prefill(self, x, return_all_logits=False):
        """x: inputs_embeds [S, D] (any float dtype, device).  Fills the KV cache for positions 0..S-1 and returns
        fp32 logits of the last position [V] (or all positions [S, V])."""
        S = x.shape[0]
        if S > self.max_seq_len:
            raise ValueError(f"sequence length {S} exceeds the KV cache ({self.max_seq_len})")
        x = x.to(device=self._dev, dtype=torch.bfloat16).contiguous()
        nh, nkv, hd, D = self.nh, self.nkv, self.hd, self.D
        q = torch.empty((S, nh * hd), dtype=torch.bfloat16, device=self._dev)
        o = torch.empty((S, nh * hd), dtype=torch.bfloat16, device=self._dev)
        smax = self.max_seq_len
        for li, lw in enumerate(self.w["layers"]):
            h = ops.rmsnorm(x, lw["ln1_w"], self.eps)
            qkv = ops.gemm(h, lw["wqkv"])
            ops.rope_kv(qkv, q, self.kcache[li], self.vcache[li], self.cos_t, self.sin_t, nh, nkv, 0)
            ops.attn_fwd(q, self.kcache[li], self.vcache[li], o, (0, hd, nh * hd), (0, smax * hd, hd), (0, smax * hd, hd),
                         (0, hd, nh * hd), 1, nh, S, S, nh // nkv, hd ** -0.5, True, 0, hd)
            x = ops.gemm(o, lw["wo"], res=x)
            h = ops.rmsnorm(x, lw["ln2_w"], self.eps)
            a = ops.gemm(h, lw["wgu"], swiglu=True)
            x = ops.gemm(a, lw["wd"], res=x)
        self.pos = S
        self.last_hidden = x
        if return_all_logits:
            h = ops.rmsnorm(x, self.w["norm_w"], self.eps)
            return ops.gemm(h, self.w["lm_head"], out_f32=True)
        return ops.gemv(self.w["lm_head"], x[S - 1], norm_w=self.w["norm_w"], eps=self.eps, out_f32=True)

    # ------------------------------------------------------------------ decode (M = 1, HBM-bound GEMVs)
    @torch.no_grad()
    def decode_step(self, tok_dev):
        """One token (device int32 [1]) at position self.pos -> fp32 logits [V]."""
        if self.pos >= self.max_seq_len:
            raise ValueError("KV cache exhausted")
        nh, nkv, hd = self.nh, self.nkv, self.hd
        x = torch.empty((1, self.D), dtype=torch.bfloat16, device=self._dev)
        ops.embed_rows(tok_dev, self.w["embed"], x)
        x = x[0]
        q = torch.empty((1, nh * hd), dtype=torch.bfloat16, device=self._dev)
        o = torch.empty((nh * hd,), dtype=torch.bfloat16, device=self._dev)
        for li, lw in enumerate(self.w["layers"]):
            qkv = ops.gemv(lw["wqkv"], x, norm_w=lw["ln1_w"], eps=self.eps)
            ops.rope_kv(qkv.view(1, -1), q, self.kcache[li], self.vcache[li], self.cos_t, self.sin_t, nh, nkv, self.pos)
            ops.attn_decode(q, self.kcache[li], self.vcache[li], self.partial, o, nh, nkv, self.pos + 1, self.decode_chunk,
                            hd ** -0.5)
            x = ops.gemv(lw["wo"], o, res=x)
            a = ops.gemv(lw["wgu"], x, norm_w=lw["ln2_w"], eps=self.eps, swiglu=True)
            x = ops.gemv(lw["wd"], a, res=x)
        self.pos += 1
        return ops.gemv(self.w["lm_head"], x, norm_w=self.w["norm_w"], eps=self.eps, out_f32=True)

    @torch.no_grad()
    def generate(self, inputs_embeds, max_new_tokens=2048, eos_token_id=None, stopping_criteria=None,
                 return_logits=False):
        """Greedy decode (HF GenerationMixin._sample, do_sample=False): returns LongTensor [1, n_new] of NEW tokens.
        Stops at `eos_token_id` (int or list), when `stopping_criteria(output_ids, None)` is truthy
        (KeywordsStoppingCriteria semantics, videollama2/mm_utils.py:341-345), or at max_new_tokens / cache end."""
        eos = set()
        if eos_token_id is not None:
            eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else {int(eos_token_id)}
        logits = self.prefill(inputs_embeds)
        hist = torch.zeros((max(max_new_tokens, 1),), dtype=torch.int32, device=self._dev)
        toks, all_logits = [], []
        for step in range(max_new_tokens):
            if return_logits:
                all_logits.append(logits.clone())
            ops.argmax(logits, self.tok, hist, step)
            t = int(self.tok.item())                 # one 4-byte D2H per token (the reference syncs per token too)
            toks.append(t)
            if t in eos:
                break
            if stopping_criteria is not None:
                ids = torch.tensor([toks], dtype=torch.long, device=self._dev)
                crit = stopping_criteria if isinstance(stopping_criteria, (list, tuple)) else [stopping_criteria]
                if any(bool(c(ids, None)) for c in crit):
                    break
            if step + 1 == max_new_tokens or self.pos >= self.max_seq_len:
                break
            logits = self.decode_step(self.tok)
        out = torch.tensor([toks], dtype=torch.long, device=self._dev)
        return (out, torch.stack(all_logits)) if return_logits else out
