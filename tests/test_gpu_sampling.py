"""do_sample=True on MI355X: csrc/k_sample.h through the C ABI against oracle/sampling_oracle.py (pinned to the live HF warpers in
tests/test_sampling.py) at the real vocabulary sizes, and generate(do_sample=True) eager == hipGraph replay under one seed."""
import os

import pytest
import torch

from oracle import sampling_oracle as SO
from oracle import vl2_oracle as O
from tests.test_sampling import CASES, check_kernel

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("V", [32000, 152064])
def test_sampled_token_on_device_against_the_oracle(V):
    from videollama2_amd import _lib, ops
    _lib.load()
    check_kernel(ops, DEV, V, cases=CASES, n_u=24)


def test_sampling_is_deterministic_and_times():
    """The same call 50 times gives one token (integer fixed-point mass: no dependence on atomic arrival order); per-call time recorded."""
    from videollama2_amd import ops
    out = {}
    for V in (32000, 152064):
        x = (torch.randn(V, generator=torch.Generator().manual_seed(3)) * 3).to(DEV)
        u = torch.full((4,), 0.37, device=DEV)
        tok = torch.zeros(1, dtype=torch.int32, device=DEV)
        seen = set()
        for _ in range(50):
            ops.sample_token(x, tok, u, 0.2, 50, 0.9)
            seen.add(int(tok.item()))
        assert len(seen) == 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name, (k, p) in (("topk50_topp0.9", (50, 0.9)), ("topp0.9_only", (0, 0.9)), ("plain", (0, 1.0))):
            e0.record()
            for _ in range(20):
                ops.sample_token(x, tok, u, 0.7, k, p)
            e1.record()
            torch.cuda.synchronize()
            out[f"V={V} {name}"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print("[sampling] us per call:", out)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    import json
    json.dump(out, open(os.path.join(d, "r05_sampling_kernel_us.json"), "w"), indent=1)


def test_generate_do_sample_graph_equals_eager_and_follows_the_oracle(golden_small):
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"], round_bf16=True), DEV, max_seq_len=64)
    ids = g["input_ids"][None].to(DEV)
    kw = dict(images=[(g["frames"].to(DEV), "video")], attention_mask=torch.ones_like(ids), max_new_tokens=8, return_logits=True,
              do_sample=True, temperature=1.2, top_k=30, top_p=0.85)
    gen = lambda: torch.Generator(device=DEV).manual_seed(5)
    eager, lg = m.generate(ids, use_graph=False, generator=gen(), **kw)
    graph, lg2 = m.generate(ids, use_graph=True, generator=gen(), **kw)
    assert eager.tolist() == graph.tolist() and torch.equal(lg, lg2)
    us = torch.rand((8,), device=DEV, generator=gen()).cpu()
    for s in range(eager.shape[1]):
        pr = SO.probs(lg[s].cpu(), 1.2, 30, 0.85)
        t = int(eager[0, s])
        assert pr[t] > 0 or t in SO.boundary_tokens(lg[s].cpu(), 1.2, 30, 0.85)
        cdf = pr.double().cumsum(0)
        lo, hi = (float(cdf[t - 1]) if t > 0 else 0.0), float(cdf[t])
        assert lo - 2e-5 <= float(us[s]) * float(cdf[-1]) <= hi + 2e-5
    greedy = m.generate(ids, images=kw["images"], attention_mask=kw["attention_mask"], max_new_tokens=8, do_sample=False)
    k1 = m.generate(ids, images=kw["images"], attention_mask=kw["attention_mask"], max_new_tokens=8, do_sample=True, temperature=0.5, top_k=1)
    assert greedy.tolist() == k1.tolist()


def test_generate_batch_do_sample_follows_the_oracle(golden_small):
    """Batched do_sample (round 6; VERDICT r05 'missing' 6): every request of `generate_batch` draws from HF's warped distribution of ITS logits at ITS
    uniform number (one ops.sample_token launch per request and step); top_k = 1 reproduces the batched greedy tokens."""
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"], round_bf16=True), DEV, max_seq_len=64)
    emb = g["inputs_embeds"].to(DEV)
    reqs = [emb, emb[:30], emb[:37]]
    T, K, P, n_new = 1.1, 20, 0.9, 6
    gen = lambda: torch.Generator(device=DEV).manual_seed(9)
    toks, lg = m.decoder.generate_batch(reqs, max_new_tokens=n_new, return_logits=True, sampler=(T, K, P, gen()))
    g2 = gen()
    n_u = min(n_new, m.decoder.max_seq_len) + 1
    us = torch.stack([torch.rand((n_u,), device=DEV, generator=g2) for _ in reqs]).cpu()
    for b in range(len(reqs)):
        assert toks[b].numel() == n_new
        for s in range(n_new):
            pr = SO.probs(lg[s, b].cpu(), T, K, P)
            t = int(toks[b][s])
            assert pr[t] > 0 or t in SO.boundary_tokens(lg[s, b].cpu(), T, K, P), (b, s)
            cdf = pr.double().cumsum(0)
            lo, hi = (float(cdf[t - 1]) if t > 0 else 0.0), float(cdf[t])
            assert lo - 2e-5 <= float(us[b, s]) * float(cdf[-1]) <= hi + 2e-5, (b, s)
    greedy = m.decoder.generate_batch(reqs, max_new_tokens=n_new)
    k1 = m.decoder.generate_batch(reqs, max_new_tokens=n_new, sampler=(0.7, 1, 1.0))
    assert [t.tolist() for t in greedy] == [t.tolist() for t in k1]
    # the model-level entry: a right-padded batch with do_sample no longer refuses
    ids = g["input_ids"][None].to(DEV)
    ids2 = torch.cat([ids, ids], 0)
    out = m.generate(ids2, images=[(g["frames"].to(DEV), "video")] * 2, attention_mask=torch.ones_like(ids2), max_new_tokens=4, do_sample=True, temperature=0.5, top_k=1)
    ref = m.generate(ids2, images=[(g["frames"].to(DEV), "video")] * 2, attention_mask=torch.ones_like(ids2), max_new_tokens=4, do_sample=False)
    assert out.tolist() == ref.tolist()
