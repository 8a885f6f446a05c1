#!/usr/bin/env python
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vl2_oracle as O
from videollama2_amd import ops
from videollama2_amd.decoder import HipMistralDecoder
for vocab, smax in ((4096, 512), (4096, 2048), (152064, 512)):
    cfg = O.config_videollama2_1_7b_16f(16)
    cfg["llm"]["num_hidden_layers"] = 2
    cfg["llm"]["vocab_size"] = vocab
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    x = (torch.randn(300, 3584, generator=torch.Generator().manual_seed(2)) * 0.5).bfloat16().float()
    dec = HipMistralDecoder(cfg, sd, "cuda", max_seq_len=smax)
    out, lg = dec.generate(x.cuda(), max_new_tokens=4, return_logits=True)
    # stage path without a graph
    dec.prefill(x.cuda())
    d, _, ws = dec._stage_desc()
    dec.state.copy_(torch.tensor([dec.pos - 1, 0], dtype=torch.int32))
    st = []
    for s in range(3):
        ops.llm_decode_step(d, dec.logits, dec.tok, dec.state, dec.hist, dec.partial, ws)
        torch.cuda.synchronize()
        st.append((int(dec.tok), torch.equal(dec.logits, lg[s + 1]), (dec.logits - lg[s + 1]).abs().max().item()))
    graph = dec.generate(x.cuda(), max_new_tokens=4, use_graph=True)
    print(vocab, smax, "eager", out[0].tolist(), "stage-eager", st, "graph", graph[0].tolist(), flush=True)
