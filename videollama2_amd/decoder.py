"""HipMistralDecoder -- the decoder backend behind `Videollama2MistralForCausalLM.generate(inputs, images=...)`
(videollama2/model/videollama2_mistral.py:110-144) and, with the q/k/v bias HF Qwen2Attention adds, behind
`Videollama2Qwen2ForCausalLM.generate` (videollama2/model/videollama2_qwen2.py:108-142; alias `HipQwen2Decoder`): consumes `inputs_embeds [S, hidden]`, runs the Mistral prefill and
the greedy decode loop on its own KV cache and returns the NEW token ids, as HF `generate` does on the
`inputs_embeds` path.  Per-layer math follows HF:models/mistral/modeling_mistral.py (RMSNorm fp32 statistics,
rotate-half RoPE theta=1e6, causal GQA attention, SwiGLU), executed by libvl2hip.so kernels only."""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, ops
from .weights import pack_decoder


class HipMistralDecoder(nn.Module):
    """tp_group: optional torch.distributed group for tensor parallelism over the heads / MLP width (the
    reference has none -- its 72B path is accelerate's device_map="auto").  Every rank holds 1/tp of the q/k/v/gate/up rows
    and of the o/down columns and of the KV cache; the two row-parallel projections of a layer produce partial sums that are
    all-reduced (RCCL over xGMI; 2 x [S, D] bf16 per layer in the prefill, 2 x [D] per layer per token in the decode loop).
    The residual is folded into rank 0's partial so the sum needs no extra pass."""

    def __init__(self, cfg, state_dict, device="cuda", max_seq_len=4096, n_layers=None, tp_group=None, tp_shard=None, tp_local=None):
        super().__init__()
        self.cfg = cfg
        l = cfg["llm"]
        self._dev = torch.device(device)
        self.tp_group = tp_group
        self.tp = dist.get_world_size(tp_group) if tp_group is not None else 1
        self.tp_rank = dist.get_rank(tp_group) if tp_group is not None else 0
        if tp_shard is not None:                 # (rank, size) WITHOUT a group: one rank's shard run alone, no collectives --
            self.tp_rank, self.tp = tp_shard     # timing model only (scripts/tp_model.py); the numbers it produces are partial sums
        # ... unless `tp_local` (dist.LocalTensorParallel) stands in for the group: every rank's shard in ONE process, the all-reduce
        # a rendezvous of host threads -- how >= 2 REAL shards are summed and checked on a 1-GPU box (tests/test_gpu_tp.py)
        self.tp_local = tp_local
        self.w = pack_decoder(state_dict, cfg, self._dev, n_layers, self.tp_rank, self.tp)
        self.n_layers = len(self.w["layers"])
        self.nh, self.nkv, self.hd = l["num_attention_heads"] // self.tp, l["num_key_value_heads"] // self.tp, l["head_dim"]
        self.D, self.V, self.eps = l["hidden_size"], l["vocab_size"], l["rms_norm_eps"]
        self.max_seq_len = max_seq_len
        # RoPE tables, fp32, HF MistralRotaryEmbedding: inv_freq = theta^(-2i/d); cos/sin of pos*inv_freq
        inv = 1.0 / (l["rope_theta"] ** (torch.arange(0, self.hd, 2, dtype=torch.int64).float() / self.hd))
        fr = torch.arange(max_seq_len, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos_t = fr.cos().contiguous().to(self._dev)
        self.sin_t = fr.sin().contiguous().to(self._dev)
        self._elem = _lib.elem()
        bf = dict(dtype=_lib.elem_dtype(), device=self._dev)
        self.kcache = [torch.zeros((self.nkv, max_seq_len, self.hd), **bf) for _ in range(self.n_layers)]
        self.vcache = [torch.zeros((self.nkv, max_seq_len, self.hd), **bf) for _ in range(self.n_layers)]
        nsplit_max = (max_seq_len + 63) // 64
        self.partial = torch.empty((self.nh * nsplit_max * 130,), dtype=torch.float32, device=self._dev)
        # decode-step state lives on the device so that one captured hipGraph replays for every token:
        #   tok = current token id, state = {position of the token being fed, step index}, hist = generated ids
        self.tok = torch.zeros((1,), dtype=torch.int32, device=self._dev)
        self.state = torch.zeros((2,), dtype=torch.int32, device=self._dev)
        self.hist = torch.zeros((max_seq_len,), dtype=torch.int32, device=self._dev)
        I = l["intermediate_size"] // self.tp
        self._b = dict(x0=torch.empty((1, self.D), **bf), qkv=torch.empty(((self.nh + 2 * self.nkv) * self.hd,), **bf),
                       o=torch.empty((self.nh * self.hd,), **bf), x1=torch.empty((self.D,), **bf),
                       a=torch.empty((I,), **bf))
        self.logits = torch.empty((self.V,), dtype=torch.float32, device=self._dev)
        self.graph = None
        self.pos = 0
        self._stage = None                  # (vl2_llm_desc, keepalive, decode workspace): prefill / decode step as ONE C call each
        # validation hook for 1-GPU boxes: issue the tensor-parallel all-reduces even when the group has ONE rank (the sum over one
        # rank is the identity), so that the captured-RCCL decode graph can be exercised on hardware without a second GPU
        self.tp_always_reduce = False

    @torch.no_grad()
    def enable_fp8_decode(self, on=True):
        """OPTIONAL arithmetic (the fp8 half of BASELINE.json configs[4]): the decode step's projections stream fp8
        (OCP e4m3fn) copies of the packed weights with one power-of-two scale per output row (csrc/k_fp8.h; activations stay 16-bit).
        Prefill keeps the 16-bit weights (it is MFMA-bound, decode is bound by the weight stream).  The copies are made once, here
        (+7.2 GB for the 7B model).  Not the reference's arithmetic and never the default: oracle/fp8_oracle.py defines the quantiser."""
        if self.tp > 1:
            raise NotImplementedError("fp8 decode weights: single-rank decoders only")
        if on and getattr(self, "w8", None) is None:
            self._make_w8()
        self.decode_fp8 = bool(on)
        self._invalidate_graphs()                                 # a captured step (greedy or sampled) holds the other projections
        return self

    def _make_w8(self):
        """The fp8 copies of the packed projections (shared by the decode and the prefill switch).  The stage descriptor is rebuilt with
        the fp8 pointers -- and with it the decode workspace, so every captured graph (which points into the old workspace) goes too."""
        self.w8 = dict(layers=[{k: ops.quant_fp8(lw[k]) for k in ("wqkv", "wo", "wgu", "wd")} for lw in self.w["layers"]],
                       lm_head=ops.quant_fp8(self.w["lm_head"]))
        self._stage = None
        self._invalidate_graphs()

    def _invalidate_graphs(self):
        """Drop every captured decode graph: the greedy one and the sampled one (keyed by the sampler only) both bake in the weight
        pointers / arithmetic (`decode_fp8`) and the stage workspace of the moment they were captured."""
        self.graph = None
        self._graph_sample = (None, None)

    @torch.no_grad()
    def enable_fp8_prefill(self, on=True):
        """OPTIONAL arithmetic (BASELINE.json configs[4] "fp8 MFMA on CDNA4"): the four projections of every layer of the
        PREFILL run on the fp8 matrix pipe (v_mfma_f32_32x32x64_f8f6f4, twice the 16-bit MFMA rate and half the operand bytes): weights = the
        row-scaled e4m3fn copies of enable_fp8_decode, activations quantised per token row on the fly (W8A8; csrc/k_fp8.h quant_act_fp8_kernel,
        k_gemm.h gemm3 / gemm4 FP8).  Attention, RoPE, the KV cache and lm_head stay 16-bit.  Not the reference's arithmetic, never the
        default: oracle/fp8_oracle.py (gemm_w8a8) defines it."""
        if self.tp > 1:
            raise NotImplementedError("fp8 prefill: single-rank decoders only")
        if on and getattr(self, "w8", None) is None:
            self._make_w8()
        self.prefill_fp8 = bool(on)
        return self

    def _stage_desc(self):
        if self._stage is None:
            d, keep = ops.llm_desc(self.w, self.cfg["llm"], self.nh, self.nkv, self.max_seq_len, self.eps, self.kcache, self.vcache,
                                   self.cos_t, self.sin_t, w8=getattr(self, "w8", None))
            ws, _ = ops._llm_ws(d, 1, self._dev)
            self._stage = (d, keep, ws)
        return self._stage

    def _use_stage(self, cache=None):
        return ops.stage_enabled() and self.tp == 1 and cache is None and not self.tp_always_reduce

    def _reduce(self, t):
        """Sum the row-parallel partial results over the tensor-parallel group (no-op without one).  gloo (CPU tests, debug)
        takes device tensors through host memory."""
        if self.tp > 1 and self.tp_local is not None:
            return self.tp_local.reduce(self.tp_rank, t)
        if (self.tp > 1 or self.tp_always_reduce) and self.tp_group is not None:
            if t.is_cuda and dist.get_backend(self.tp_group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, group=self.tp_group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.tp_group)
        return t

    def _row_parallel(self, a, w, x, rs, rn):
        """x + a @ w.T (o_proj / down_proj with the residual fused); leaves (0, rstd) of the result's rows in `rn` for the next
        norm-carrying GEMM.  Single rank: the GEMM's epilogue emits the partial statistics (`rs`).  Tensor parallel: the GEMM output
        is a partial sum, so the statistics are taken after the all-reduce (one extra read of x)."""
        if self.tp > 1:
            x = self._reduce(ops.gemm(a, w, res=x if self.tp_rank == 0 else None))
            ops.row_stats(x, out=rs)
        else:
            x = ops.gemm(a, w, res=x, stats_out=rs)
        ops.row_norm_finalize(rs, self.D, ops.NORM_RMS, self.eps, out=rn)
        return x

    # ------------------------------------------------------------------ prefill (M = S tokens, MFMA GEMMs)
    @torch.no_grad()
    def prefill(self, x, return_all_logits=False, cache=None, logits_out=None):
        """x: inputs_embeds [S, D] (any float dtype, device).  Fills the KV cache for positions 0..S-1 and returns
        fp32 logits of the last position [V] (or all positions [S, V]).  cache = (k per layer, v per layer) overrides the
        decoder's own single-sequence cache (batched decode gives every sequence its slice)."""
        _lib.check_elem(self._elem, type(self).__name__)
        kcache, vcache = cache if cache is not None else (self.kcache, self.vcache)
        S = x.shape[0]
        if S > self.max_seq_len:
            raise ValueError(f"sequence length {S} exceeds the KV cache ({self.max_seq_len})")
        x = x.to(device=self._dev, dtype=_lib.elem_dtype()).contiguous()
        if self._use_stage(cache) and not return_all_logits:      # the whole prefill as one call into libvl2hip.so (vl2_llm_prefill)
            out = self.logits if logits_out is None else logits_out
            ops.llm_prefill(self._stage_desc()[0], x, out, fp8=getattr(self, "prefill_fp8", False))
            self.pos = S
            self.last_hidden = None
            return out
        nh, nkv, hd, D = self.nh, self.nkv, self.hd, self.D
        q = torch.empty((S, nh * hd), dtype=_lib.elem_dtype(), device=self._dev)
        o = torch.empty((S, nh * hd), dtype=_lib.elem_dtype(), device=self._dev)
        smax = self.max_seq_len
        if getattr(self, "prefill_fp8", False):                   # the same pass operator by operator on the fp8 matrix pipe (vl2_stage.inc)
            for li, (lw, w8) in enumerate(zip(self.w["layers"], self.w8["layers"])):
                a8, tab = ops.quant_act_fp8(x, rms_eps=self.eps)
                qkv = ops.gemm_fp8(a8, tab, *w8["wqkv"], bias=lw["bqkv"])
                ops.rope_kv(qkv, q, kcache[li], vcache[li], self.cos_t, self.sin_t, nh, nkv, 0)
                ops.attn_fwd(q, kcache[li], vcache[li], o, (0, hd, nh * hd), (0, smax * hd, hd), (0, smax * hd, hd),
                             (0, hd, nh * hd), 1, nh, S, S, nh // nkv, hd ** -0.5, True, 0, hd)
                a8, tab = ops.quant_act_fp8(o)
                x1 = ops.gemm_fp8(a8, tab, *w8["wo"], res=x)
                a8, tab = ops.quant_act_fp8(x1, rms_eps=self.eps)
                a = ops.gemm_fp8(a8, tab, *w8["wgu"], swiglu=True)
                a8, tab = ops.quant_act_fp8(a)
                x = ops.gemm_fp8(a8, tab, *w8["wd"], res=x1)
            self.pos = S
            self.last_hidden = x
            if return_all_logits:
                h = ops.rmsnorm(x, self.w["norm_w"], self.eps)
                return ops.gemm(h, self.w["lm_head"], out_f32=True)
            return ops.gemv(self.w["lm_head"], x[S - 1], norm_w=self.w["norm_w"], eps=self.eps, out_f32=True,
                            out=self.logits if logits_out is None else logits_out)
        rs = ops.row_stats(x)          # RMSNorm rides in the q/k/v and gate/up GEMMs (weights.fold_norm): this seeds the statistics
        rn = ops.row_norm_finalize(rs, D, ops.NORM_RMS, self.eps)       # [S, 2] (0, rstd): reduced once, not in every column tile
        for li, lw in enumerate(self.w["layers"]):
            qkv = ops.gemm(x, lw["wqkv"], bias=lw["bqkv"], norm=(ops.NORM_RMS, rn, self.eps, None))   # bqkv: Qwen2 only
            ops.rope_kv(qkv, q, kcache[li], vcache[li], self.cos_t, self.sin_t, nh, nkv, 0)
            ops.attn_fwd(q, kcache[li], vcache[li], o, (0, hd, nh * hd), (0, smax * hd, hd), (0, smax * hd, hd),
                         (0, hd, nh * hd), 1, nh, S, S, nh // nkv, hd ** -0.5, True, 0, hd)
            x = self._row_parallel(o, lw["wo"], x, rs, rn)
            a = ops.gemm(x, lw["wgu"], swiglu=True, norm=(ops.NORM_RMS, rn, self.eps, None), mfma16=True)
            x = self._row_parallel(a, lw["wd"], x, rs, rn)
        self.pos = S
        self.last_hidden = x
        if return_all_logits:
            h = ops.rmsnorm(x, self.w["norm_w"], self.eps)
            return ops.gemm(h, self.w["lm_head"], out_f32=True)
        return ops.gemv(self.w["lm_head"], x[S - 1], norm_w=self.w["norm_w"], eps=self.eps, out_f32=True,
                        out=self.logits if logits_out is None else logits_out)

    # ------------------------------------------------------------------ decode (M = 1, HBM-bound GEMVs)
    def _decode_kernels(self, dyn):
        """Enqueue one decode step for the token in self.tok: embed -> 32 x {qkv GEMV (+RMSNorm), RoPE+append+attention,
        o GEMV (+res), gate/up GEMV (+RMSNorm, SwiGLU), down GEMV (+res)} -> lm_head GEMV (+final RMSNorm) into self.logits.
        dyn=True reads the position from self.state[0] on the device (hipGraph-replayable); no allocation either way."""
        b, nh, nkv, hd = self._b, self.nh, self.nkv, self.hd
        pos_dev = self.state[0:1] if dyn else None
        ops.embed_rows(self.tok, self.w["embed"], b["x0"])
        x = b["x0"][0]
        if getattr(self, "decode_fp8", False):                  # the same step on the fp8 copies (vl2_gemv_fp8), operator by operator
            for li, (lw, q) in enumerate(zip(self.w["layers"], self.w8["layers"])):
                ops.gemv_fp8(*q["wqkv"], x, eps=self.eps, out=b["qkv"], bias=lw["bqkv"], rms_plain=True)
                ops.attn_decode(b["qkv"], self.kcache[li], self.vcache[li], self.cos_t, self.sin_t, self.partial, b["o"], nh, nkv,
                                self.pos, hd ** -0.5, pos_dev=pos_dev, ctx_cap=self.max_seq_len)
                ops.gemv_fp8(*q["wo"], b["o"], res=x, out=b["x1"])
                ops.gemv_fp8(*q["wgu"], b["x1"], eps=self.eps, swiglu=True, out=b["a"], rms_plain=True)
                ops.gemv_fp8(*q["wd"], b["a"], res=b["x1"], out=x)
            ops.gemv_fp8(*self.w8["lm_head"], x, norm_w=self.w["norm_w"], eps=self.eps, out_f32=True, out=self.logits)
            return
        for li, lw in enumerate(self.w["layers"]):
            ops.gemv(lw["wqkv"], x, norm_w=self.w["ones"], eps=self.eps, out=b["qkv"], bias=lw["bqkv"])   # ln weight folded into wqkv
            ops.attn_decode(b["qkv"], self.kcache[li], self.vcache[li], self.cos_t, self.sin_t, self.partial, b["o"], nh, nkv,
                            self.pos, hd ** -0.5, pos_dev=pos_dev, ctx_cap=self.max_seq_len)
            r0 = self.tp_rank == 0                                              # the residual rides on rank 0's partial sum
            self._reduce(ops.gemv(lw["wo"], b["o"], res=x if r0 else None, out=b["x1"]))          # x1 = x + attn
            ops.gemv(lw["wgu"], b["x1"], norm_w=self.w["ones"], eps=self.eps, swiglu=True, out=b["a"])
            self._reduce(ops.gemv(lw["wd"], b["a"], res=b["x1"] if r0 else None, out=x))          # x = x1 + mlp (x's old value is dead)
        ops.gemv(self.w["lm_head"], x, norm_w=self.w["norm_w"], eps=self.eps, out_f32=True, out=self.logits)

    @torch.no_grad()
    def decode_step(self, tok_dev=None):
        """Eager step: feed the token in self.tok (or tok_dev) at position self.pos -> fp32 logits [V] (self.logits)."""
        _lib.check_elem(self._elem, type(self).__name__)
        if self.pos >= self.max_seq_len:
            raise ValueError("KV cache exhausted")
        if tok_dev is not None and tok_dev.data_ptr() != self.tok.data_ptr():
            self.tok.copy_(tok_dev)
        self._decode_kernels(dyn=False)
        self.pos += 1
        return self.logits

    @torch.no_grad()
    def capture_graph(self, sampler=None):
        """sampler = (temperature, top_k, top_p): the sampled-token launch (ops.sample_token, reading its uniform number u[step] from
        self.u_buf on the device) takes the argmax's place in the captured step; one graph per sampler setting.
        Capture {argmax -> decode step} once as a hipGraph (torch.cuda.CUDAGraph records the launches libvl2hip.so
        enqueues on the capture stream).  Replays read token / position / step from device memory.
        Tensor-parallel decoders: the two all-reduces per layer are RCCL kernels on the capture stream and become graph nodes
        like every other launch (ProcessGroupNCCL supports stream capture); the host-staged gloo debug path cannot be captured."""
        if (self.tp > 1 or self.tp_always_reduce) and (self.tp_group is None or dist.get_backend(self.tp_group) != "nccl"):
            raise NotImplementedError("hipGraph decode under tensor parallelism needs the nccl (RCCL) backend: gloo stages through the host")
        if sampler is None and self.graph is not None:
            return self.graph
        if sampler is not None and getattr(self, "_graph_sample", (None, None))[0] == tuple(sampler):
            return self._graph_sample[1]
        saved = (self.state.clone(), self.tok.clone(), self.logits.clone(), self.hist[:2].clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        def one_step():                                       # argmax + the token's forward: ONE call into libvl2hip.so
            if sampler is not None:                           # (the stage call opens with its own argmax: the sampled step is the operator sequence)
                ops.sample_token(self.logits, self.tok, self.u_buf, sampler[0], sampler[1], sampler[2], hist=self.hist, state=self.state)
                self._decode_kernels(dyn=True)
            elif self._use_stage():
                d, _, ws = self._stage_desc()
                ops.llm_decode_step(d, self.logits, self.tok, self.state, self.hist, self.partial, ws, fp8=getattr(self, "decode_fp8", False))
            else:
                ops.argmax(self.logits, self.tok, self.hist, 0, self.state)
                self._decode_kernels(dyn=True)

        with torch.cuda.stream(side):                       # warm-up outside capture (first-launch attribute calls etc.)
            self.state.copy_(torch.tensor([max(self.pos - 1, 0), 0], dtype=torch.int32))
            one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # a RCCL watchdog thread may be polling events
            one_step()
        self.state.copy_(saved[0]); self.tok.copy_(saved[1]); self.logits.copy_(saved[2]); self.hist[:2].copy_(saved[3])
        torch.cuda.synchronize()
        if sampler is not None:
            self._graph_sample = (tuple(sampler), g)
        else:
            self.graph = g
        return g

    @torch.no_grad()
    def generate(self, inputs_embeds, max_new_tokens=2048, eos_token_id=None, stopping_criteria=None,
                 return_logits=False, use_graph=False, streamer=None, sampler=None):
        """Greedy decode (HF GenerationMixin._sample, do_sample=False): returns LongTensor [1, n_new] of NEW tokens.
        Stops at `eos_token_id` (int or list), when `stopping_criteria(output_ids, None)` is truthy
        (KeywordsStoppingCriteria semantics, videollama2/mm_utils.py:341-345), or at max_new_tokens / cache end.
        use_graph=True replays one captured hipGraph per token (argmax + the whole decode step).
        streamer: optional object with put(LongTensor[1, n]) / end() (the HF `BaseStreamer` protocol the reference's worker
        uses with TextIteratorStreamer, serve/model_worker.py:263-300): every new token is handed over as soon as it is known.
        sampler = (temperature, top_k, top_p[, generator]): HF `_sample` with do_sample=True -- the logits warpers in HF's order and one draw per
        step (ops.sample_token, csrc/k_sample.h) instead of the argmax; the uniform numbers come from torch's generator for this device (or the
        given one), so `torch.manual_seed` makes a run repeatable like it does for the reference."""
        eos = set()
        if eos_token_id is not None:
            eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else {int(eos_token_id)}
        logits = self.prefill(inputs_embeds)
        if logits.data_ptr() != self.logits.data_ptr():
            self.logits.copy_(logits)
        max_new_tokens = min(max_new_tokens, self.max_seq_len - self.pos + 1)
        crit = None
        if stopping_criteria is not None:
            crit = stopping_criteria if isinstance(stopping_criteria, (list, tuple)) else [stopping_criteria]
        toks, all_logits = [], []
        # a prompt that fills the cache leaves no position to decode into: the capture warm-up would run a step at row
        # max_seq_len (the kernels now ignore such a step, but there is nothing to replay either) -> plain last-step path
        tp_ok = self.tp == 1 or (self.tp_group is not None and dist.get_backend(self.tp_group) == "nccl")
        use_graph = use_graph and tp_ok and self._dev.type == "cuda" and self.pos < self.max_seq_len
        if sampler is not None:
            T, tk, tp = float(sampler[0]), int(sampler[1]), float(sampler[2])
            gen = sampler[3] if len(sampler) > 3 else None
            if getattr(self, "u_buf", None) is None:
                self.u_buf = torch.zeros((self.max_seq_len + 1,), dtype=torch.float32, device=self._dev)
            self.u_buf[:max_new_tokens].copy_(torch.rand((max_new_tokens,), device=self._dev, generator=gen))
            sampler = (T, tk, tp)
        if use_graph:
            g = self.capture_graph(sampler)
            self.state.copy_(torch.tensor([self.pos - 1, 0], dtype=torch.int32))
        for step in range(max_new_tokens):
            if return_logits:
                all_logits.append(self.logits.clone())
            last = step + 1 == max_new_tokens or self.pos >= self.max_seq_len
            if use_graph and not last:
                g.replay()                           # argmax(step) + forward of the new token -> logits(step+1)
                self.pos += 1
            elif sampler is not None:
                ops.sample_token(self.logits, self.tok, self.u_buf, sampler[0], sampler[1], sampler[2], hist=self.hist, step=step)
            else:
                ops.argmax(self.logits, self.tok, self.hist, step)
            t = int(self.tok.item())                 # one 4-byte D2H per token (the reference syncs per token too)
            toks.append(t)
            if streamer is not None:
                streamer.put(torch.tensor([[t]], dtype=torch.long))
            if t in eos or last:
                break
            if crit is not None:
                ids = torch.tensor([toks], dtype=torch.long, device=self._dev)
                if any(bool(c(ids, None)) for c in crit):
                    break
            if not use_graph:
                self.decode_step()
        if streamer is not None:
            streamer.end()
        out = torch.tensor([toks], dtype=torch.long, device=self._dev)
        return (out, torch.stack(all_logits)) if return_logits else out

    # ------------------------------------------------------------------ batched prefill / decode
    @torch.no_grad()
    def prefill_batch(self, xs, caches, logits_out):
        """Several prompts prefilled together: the projections and norms run once on the concatenated rows [sum S_b, D] (bigger
        M: fuller GEMM grids), RoPE / cache fill / causal attention per sequence on its own cache.  Every kernel here is
        row-independent, so the result is bit-identical to prefilling the prompts one by one.
        xs: list of [S_b, D]; caches: list of (k per layer, v per layer); logits_out: [B, V] fp32 (last position of each)."""
        if getattr(self, "prefill_fp8", False):
            raise NotImplementedError("fp8 prefill (enable_fp8_prefill) covers the single-sequence prefill; the batched prefill runs the 16-bit "
                                      "projections: call enable_fp8_prefill(False) first")
        lens = [x.shape[0] for x in xs]
        if max(lens) > self.max_seq_len:
            raise ValueError(f"sequence length {max(lens)} exceeds the KV cache ({self.max_seq_len})")
        X = torch.cat([x.to(device=self._dev, dtype=_lib.elem_dtype()) for x in xs], 0).contiguous()
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        nh, nkv, hd, smax = self.nh, self.nkv, self.hd, self.max_seq_len
        q = torch.empty((offs[-1], nh * hd), dtype=_lib.elem_dtype(), device=self._dev)
        o = torch.empty((offs[-1], nh * hd), dtype=_lib.elem_dtype(), device=self._dev)
        rs = ops.row_stats(X)
        rn = ops.row_norm_finalize(rs, self.D, ops.NORM_RMS, self.eps)
        for li, lw in enumerate(self.w["layers"]):
            qkv = ops.gemm(X, lw["wqkv"], bias=lw["bqkv"], norm=(ops.NORM_RMS, rn, self.eps, None))
            for b, (kc, vc) in enumerate(caches):
                s0, s1, S = offs[b], offs[b + 1], lens[b]
                ops.rope_kv(qkv[s0:s1], q[s0:s1], kc[li], vc[li], self.cos_t, self.sin_t, nh, nkv, 0)
                ops.attn_fwd(q[s0:s1], kc[li], vc[li], o[s0:s1], (0, hd, nh * hd), (0, smax * hd, hd), (0, smax * hd, hd),
                             (0, hd, nh * hd), 1, nh, S, S, nh // nkv, hd ** -0.5, True, 0, hd)
            X = self._row_parallel(o, lw["wo"], X, rs, rn)
            a = ops.gemm(X, lw["wgu"], swiglu=True, norm=(ops.NORM_RMS, rn, self.eps, None), mfma16=True)
            X = self._row_parallel(a, lw["wd"], X, rs, rn)
        for b in range(len(xs)):
            ops.gemv(self.w["lm_head"], X[offs[b + 1] - 1], norm_w=self.w["norm_w"], eps=self.eps, out_f32=True, out=logits_out[b])
        return lens

    def _ensure_batch(self, B, owner=None):
        """Slot buffers (KV caches, positions, tokens, logits) of the batched decode paths.  They are SHARED by `generate_batch`,
        `model.generate(batch > 1)` and `serving.ContinuousBatcher`; a batcher with requests in flight marks them busy
        (`_bb_busy` = the batcher) and every other user is refused until it drains, instead of silently overwriting the caches and
        positions of the requests in flight."""
        busy = getattr(self, "_bb_busy", None)
        if busy is not None and busy is not owner and busy.in_flight():
            raise RuntimeError("the decoder's batch slots are in use by a ContinuousBatcher with requests in flight: drain it "
                               "(run()) or use a second decoder -- generate_batch / batched generate / another batcher would "
                               "overwrite their KV caches")
        if getattr(self, "_bb", None) is not None and self._bb["B"] >= B:
            return self._bb
        self._batch_graphs = {}                      # captured graphs point into the buffers replaced below
        if self._dev.type == "cuda":
            ops.attach_workspace(self._dev)          # fp32 partial sums of the skinny-M GEMMs of a large-batch decode step
        bf = dict(dtype=_lib.elem_dtype(), device=self._dev)
        smax, I = self.max_seq_len, self.cfg["llm"]["intermediate_size"] // self.tp
        self._bb = dict(
            B=B, k=[torch.zeros((B, self.nkv, smax, self.hd), **bf) for _ in range(self.n_layers)],
            v=[torch.zeros((B, self.nkv, smax, self.hd), **bf) for _ in range(self.n_layers)],
            partial=torch.empty((B * self.nh * ((smax + 63) // 64) * 130,), dtype=torch.float32, device=self._dev),
            x0=torch.empty((B, self.D), **bf), x1=torch.empty((B, self.D), **bf), qkv=torch.empty((B, (self.nh + 2 * self.nkv) * self.hd), **bf),
            o=torch.empty((B, self.nh * self.hd), **bf), a=torch.empty((B, I), **bf),
            logits=torch.empty((B, self.V), dtype=torch.float32, device=self._dev),
            tok=torch.zeros((B,), dtype=torch.int32, device=self._dev), pos=torch.zeros((B,), dtype=torch.int32, device=self._dev))
        return self._bb

    PREFILL_GROUP_TOKENS = 8192   # rows prefilled in one pass (activation scratch: ~0.6 GB at 7B widths)
    GEMM_BATCH = 5      # from this many sequences on, the decode step runs its projections on MFMA (M = sequences)

    def _decode_kernels_batched(self, nb):
        """One decode step for the nb sequences of the batch: the weights stream ONCE for nb tokens, the attention runs per
        sequence on its own cache slice and position (one launch for all of them).
        nb < GEMM_BATCH: multi-row GEMV (a row is bit-identical to the single-sequence step).  nb >= GEMM_BATCH: the
        projections run on the skinny-M MFMA kernel (weights streamed once, GEMV-style, for up to 64 rows; the tiled GEMMs
        beyond that; RMSNorm as its own kernel) -- prefill-style arithmetic, i.e. equal to the single-sequence step to bf16
        rounding, not to the bit."""
        if getattr(self, "decode_fp8", False):
            raise NotImplementedError("fp8 decode weights (enable_fp8_decode) cover the single-sequence decode step; batched decode streams "
                                      "the 16-bit weights: call enable_fp8_decode(False) first")
        bb, nh, nkv, hd = self._bb, self.nh, self.nkv, self.hd
        x, x1, qkv, o, a = bb["x0"][:nb], bb["x1"][:nb], bb["qkv"][:nb], bb["o"][:nb], bb["a"][:nb]
        ops.embed_rows(bb["tok"][:nb], self.w["embed"], x)
        r0 = self.tp_rank == 0
        if nb >= self.GEMM_BATCH:
            # M = nb rows: every projection is a few-tile grid, i.e. bound by how many workgroups stream the weights; split-K
            # (partials through the workspace) is wired but off, see SPLITK_IN_BATCHED_GEMM.
            ops.set_splitk(self.SPLITK_IN_BATCHED_GEMM and self._dev.type == "cuda")
            try:
                self._batched_gemm_step(nb)
            finally:
                ops.set_splitk(False)
            return
        self._batched_gemv_step(nb)

    SPLITK_IN_BATCHED_GEMM = False      # measured: no gain (B=16 8.09 vs 8.01 ms per step): the fp32 partial exchange costs what it saves

    def _batched_gemm_step(self, nb):
        bb, nh, nkv, hd = self._bb, self.nh, self.nkv, self.hd
        x, x1, qkv, o, a = bb["x0"][:nb], bb["x1"][:nb], bb["qkv"][:nb], bb["o"][:nb], bb["a"][:nb]
        r0 = self.tp_rank == 0
        # up to 64 rows: the skinny-M kernel streams the weights GEMV-style into MFMA; beyond that the tiled GEMMs
        mm = ops.gemm_skinny if nb <= 64 else ops.gemm
        for li, lw in enumerate(self.w["layers"]):
            h = ops.rmsnorm(x, self.w["ones"], self.eps)                  # the norm weights are folded into wqkv / wgu
            mm(h, lw["wqkv"], bias=lw["bqkv"], out=qkv)
            ops.attn_decode_batched(qkv, bb["k"][li][:nb], bb["v"][li][:nb], self.cos_t, self.sin_t, bb["partial"], o, nh, nkv,
                                    bb["pos"][:nb], self.max_seq_len, hd ** -0.5)
            self._reduce(mm(o, lw["wo"], res=x if r0 else None, out=x1))
            h = ops.rmsnorm(x1, self.w["ones"], self.eps)
            mm(h, lw["wgu"], swiglu=True, out=a)
            self._reduce(mm(a, lw["wd"], res=x1 if r0 else None, out=x))
        h = ops.rmsnorm(x, self.w["norm_w"], self.eps)
        mm(h, self.w["lm_head"], out_f32=True, out=bb["logits"][:nb])
        bb["pos"][:nb] += 1

    def _batched_gemv_step(self, nb):
        bb, nh, nkv, hd = self._bb, self.nh, self.nkv, self.hd
        x, x1, qkv, o, a = bb["x0"][:nb], bb["x1"][:nb], bb["qkv"][:nb], bb["o"][:nb], bb["a"][:nb]
        r0 = self.tp_rank == 0
        for li, lw in enumerate(self.w["layers"]):
            ops.gemv_batched(lw["wqkv"], x, norm_w=self.w["ones"], eps=self.eps, out=qkv, bias=lw["bqkv"])
            ops.attn_decode_batched(qkv, bb["k"][li][:nb], bb["v"][li][:nb], self.cos_t, self.sin_t, bb["partial"], o, nh, nkv,
                                    bb["pos"][:nb], self.max_seq_len, hd ** -0.5)
            self._reduce(ops.gemv_batched(lw["wo"], o, res=x if r0 else None, out=x1))
            ops.gemv_batched(lw["wgu"], x1, norm_w=self.w["ones"], eps=self.eps, swiglu=True, out=a)
            self._reduce(ops.gemv_batched(lw["wd"], a, res=x1 if r0 else None, out=x))
        ops.gemv_batched(self.w["lm_head"], x, norm_w=self.w["norm_w"], eps=self.eps, out_f32=True, out=bb["logits"][:nb])
        bb["pos"][:nb] += 1

    @torch.no_grad()
    def _batched_step(self, nb):
        bb = self._bb
        for b in range(nb):
            ops.argmax(bb["logits"][b], bb["tok"][b:b + 1])
        self._decode_kernels_batched(nb)

    def capture_batch_graph(self, nb):
        """{argmax per request + the whole batched decode step} as ONE hipGraph per batch size (tokens and positions live on
        the device, so the same graph replays for every step)."""
        graphs = self.__dict__.setdefault("_batch_graphs", {})
        if nb in graphs:
            return graphs[nb]
        if self.tp > 1:
            raise NotImplementedError("hipGraph decode is built for the single-GPU decoder (collectives are launched eagerly)")
        bb = self._bb
        saved = (bb["tok"].clone(), bb["pos"].clone(), bb["logits"].clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up outside capture (first-launch attribute calls, allocator)
            self._batched_step(nb)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bb["pos"].copy_(saved[1])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._batched_step(nb)
        bb["tok"].copy_(saved[0]); bb["pos"].copy_(saved[1]); bb["logits"].copy_(saved[2])
        torch.cuda.synchronize()
        graphs[nb] = g
        return g

    @torch.no_grad()
    def generate_batch(self, inputs_embeds_list, max_new_tokens=2048, eos_token_id=None, return_logits=False, use_graph=None, sampler=None):
        """Greedy decode of several requests at once (not in the reference, whose eval loops run batch 1 and whose worker
        serialises requests; its padded-batch `prepare_inputs_labels_for_multimodal`, arch.py:227-261, is the nearest thing):
        every request is prefilled on its own (its M is already large), then ALL of them decode together, one token per
        request per step.  Prompts may have different lengths (per-sequence positions, no padding).  Returns a list of
        LongTensor [n_new_b] (each ends at its EOS / max_new_tokens); with return_logits also the per-step fp32 logits
        [steps, B, V].  A row of a batched step is bit-identical to the single-sequence step while nb < GEMM_BATCH.
        use_graph (default: on a GPU without tensor parallelism) replays one captured hipGraph per step.
        sampler = (temperature, top_k, top_p[, generator]): HF `_sample` with do_sample=True for every request -- one ops.sample_token launch per request
        and step in place of its argmax (request b draws from its own row u[b, step] of uniform numbers, generated request by request from the
        generator, so request 0's stream is the one a single-sequence `generate` with the same seed would use); the steps run eagerly."""
        nb = len(inputs_embeds_list)
        if use_graph is None:
            use_graph = self._dev.type == "cuda" and self.tp == 1
        if sampler is not None:
            use_graph = False
            T_, tk_, tp_ = float(sampler[0]), int(sampler[1]), float(sampler[2])
            gen = sampler[3] if len(sampler) > 3 else None
            n_u = min(max_new_tokens, self.max_seq_len) + 1
            u_rows = torch.stack([torch.rand((n_u,), device=self._dev, generator=gen) for _ in range(nb)]).contiguous()
        eos = set()
        if eos_token_id is not None:
            eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else {int(eos_token_id)}
        bb = self._ensure_batch(nb)
        # prompts are prefilled together in groups of <= PREFILL_GROUP_TOKENS rows (bit-identical to one by one)
        lens, group, gtok = [], [], 0
        caches = [([k[b] for k in bb["k"]], [v[b] for v in bb["v"]]) for b in range(nb)]

        def flush():
            if group:
                b0 = group[0]
                self.prefill_batch([inputs_embeds_list[b] for b in group], [caches[b] for b in group], bb["logits"][b0:b0 + len(group)])
                group.clear()

        for b, xb in enumerate(inputs_embeds_list):
            if gtok + xb.shape[0] > self.PREFILL_GROUP_TOKENS:
                flush()
                gtok = 0
            group.append(b)
            gtok += xb.shape[0]
            lens.append(xb.shape[0])
        flush()
        bb["pos"][:nb].copy_(torch.tensor(lens, dtype=torch.int32))
        max_new_tokens = min(max_new_tokens, self.max_seq_len - max(lens) + 1)
        outs, done, all_logits = [[] for _ in range(nb)], [False] * nb, []
        use_graph = use_graph and max(lens) < self.max_seq_len          # a full cache leaves nothing to replay (see generate)
        graph = self.capture_batch_graph(nb) if use_graph else None
        if graph is not None:                                                # capture clobbered nothing: state was restored
            bb["pos"][:nb].copy_(torch.tensor(lens, dtype=torch.int32))
        for step in range(max_new_tokens):
            if return_logits:
                all_logits.append(bb["logits"][:nb].clone())
            last = step + 1 == max_new_tokens
            if graph is not None and not last:
                graph.replay()                                               # argmax(step) + forward of the new tokens
            elif sampler is not None:
                for b in range(nb):
                    ops.sample_token(bb["logits"][b], bb["tok"][b:b + 1], u_rows[b], T_, tk_, tp_, step=step)
            else:
                for b in range(nb):
                    ops.argmax(bb["logits"][b], bb["tok"][b:b + 1])
            toks = bb["tok"][:nb].tolist()                                   # one small D2H per step for the stop checks
            for b, t in enumerate(toks):
                if not done[b]:
                    outs[b].append(t)
                    done[b] = t in eos
            if all(done) or last:
                break
            if graph is None:
                self._decode_kernels_batched(nb)
        res = [torch.tensor(o, dtype=torch.long, device=self._dev) for o in outs]
        return (res, torch.stack(all_logits)) if return_logits else res


HipQwen2Decoder = HipMistralDecoder      # same decoder; the q/k/v bias is picked up from the state dict (weights.pack_decoder)
