#!/usr/bin/env python
"""Probe: stage-level decode step (fused attention + combine) vs the per-operator step, bit by bit, at full width."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vl2_oracle as O
from videollama2_amd import ops
from videollama2_amd.decoder import HipMistralDecoder
for fam, S in (("mistral", 300), ("mistral", 1650), ("qwen2", 300)):
    cfg = O.config_videollama2_7b(16) if fam == "mistral" else O.config_videollama2_1_7b_16f(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    D = cfg["llm"]["hidden_size"]
    x = (torch.randn(S, D, generator=torch.Generator().manual_seed(2)) * 0.5).bfloat16().float()
    dec = HipMistralDecoder(cfg, sd, "cuda", max_seq_len=2048)
    l0 = dec.prefill(x.cuda()).clone()
    eager = []
    for s in range(6):
        ops.argmax(dec.logits, dec.tok)
        eager.append((int(dec.tok), dec.decode_step().clone()))
    dec2 = HipMistralDecoder(cfg, sd, "cuda", max_seq_len=2048)
    dec2.prefill(x.cuda())
    d, _, ws = dec2._stage_desc()
    dec2.state.copy_(torch.tensor([dec2.pos - 1, 0], dtype=torch.int32))
    for s in range(6):
        ops.llm_decode_step(d, dec2.logits, dec2.tok, dec2.state, dec2.hist, dec2.partial, ws)
        torch.cuda.synchronize()
        same = torch.equal(dec2.logits, eager[s][1])
        print(fam, S, "step", s, "tok", int(dec2.tok), eager[s][0], "logits equal:", same, "max|d|", (dec2.logits - eager[s][1]).abs().max().item(), flush=True)
