"""do_sample=True (videollama2/__init__.py:93-106 -> HF GenerationMixin._sample): the sampled-token kernel (csrc/k_sample.h) against
oracle/sampling_oracle.py, and the oracle against the LIVE HF logits warpers of the installed transformers.  CPU: the kernel runs on the host
emulator (tests/emu); the `-m gpu` twin is tests/test_gpu_sampling.py."""
import pytest
import torch

from oracle import sampling_oracle as SO
from tests.emu.backend import emulated_backend

CASES = [(0.2, 50, 0.9), (1.0, 50, 0.9), (0.7, 0, 0.8), (1.3, 5, 1.0), (1.0, 0, 1.0), (0.05, 50, 0.5), (2.0, 1000, 0.95)]


def _logits(V, seed, spread=4.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(V, generator=g) * spread
    x[torch.randint(0, V, (3,), generator=g)] += 6.0            # a few clear favourites, as a trained lm_head has
    return x


def test_oracle_is_the_live_hf_warper_chain():
    """Pin: the restated warpers == transformers' own classes applied in HF's order, element for element (incl. the -inf pattern)."""
    lp = pytest.importorskip("transformers.generation.logits_process")
    for seed, (T, k, p) in enumerate(CASES):
        x = _logits(3000, seed)
        s = x[None].clone()
        if T != 1.0:
            s = lp.TemperatureLogitsWarper(T)(None, s)
        if k:
            s = lp.TopKLogitsWarper(top_k=k, min_tokens_to_keep=1)(None, s)
        if p < 1.0:
            s = lp.TopPLogitsWarper(top_p=p, min_tokens_to_keep=1)(None, s)
        assert torch.equal(SO.warp(x, T, k, p), s[0]), (T, k, p)


def check_kernel(ops, dev, V, cases=CASES, n_u=24):
    for seed, (T, k, p) in enumerate(cases):
        x = _logits(V, 100 + seed)
        pr = SO.probs(x, T, k, p)
        kept = set(torch.nonzero(pr > 0).flatten().tolist())
        edge = SO.boundary_tokens(x, T, k, p)
        us = torch.rand(n_u, generator=torch.Generator().manual_seed(seed)).float()
        us[0], us[1] = 0.0, 0.99999994
        xd, ud = x.to(dev), us.to(dev)
        tok = torch.zeros(1, dtype=torch.int32, device=dev)
        hist = torch.full((n_u,), -1, dtype=torch.int32, device=dev)
        dbg = torch.zeros(4, device=dev)
        cdf = pr.double().cumsum(0)
        for i in range(n_u):
            ops.sample_token(xd, tok, ud, T, k, p, hist=hist, step=i, dbg=dbg)
            t = int(tok.item())
            assert t in kept or t in edge, f"T={T} k={k} p={p}: token {t} is outside the warpers' kept set"
            n_kept = int(dbg[0].item())
            assert abs(n_kept - len(kept)) <= len(edge), (T, k, p, n_kept, len(kept), len(edge))
            if not edge:
                # the token's interval of the oracle's cumulative distribution must contain u (to the fixed-point / exp rounding of the kernel)
                lo, hi = (float(cdf[t - 1]) if t > 0 else 0.0), float(cdf[t])
                # cumulative probability mass strictly below token t among KEPT tokens in index order == cdf[t-1]
                assert lo - 2e-5 <= float(us[i]) * float(cdf[-1]) <= hi + 2e-5, (T, k, p, t, float(us[i]), lo, hi)
        assert hist.tolist() == [int(h) for h in hist.tolist()] and int(hist[-1]) == int(tok.item())
    # state protocol (hipGraph-replayable): the step index comes from state[1]; position and step advance
    state = torch.tensor([7, 2], dtype=torch.int32, device=dev)
    ops.sample_token(xd, tok, ud, 1.0, 0, 1.0, hist=hist, state=state)
    assert state.tolist() == [8, 3] and int(hist[2]) == int(tok.item())
    # greedy limit: top_k = 1 is the argmax whatever u
    ops.sample_token(xd, tok, ud, 0.7, 1, 1.0, step=5)
    assert int(tok.item()) == int(x.argmax())


def test_emu_sampled_token_against_the_oracle():
    with emulated_backend():
        from videollama2_amd import ops
        check_kernel(ops, "cpu", 1500, n_u=6)
        check_kernel(ops, "cpu", 5000, cases=CASES[:2], n_u=4)


def test_emu_sampling_frequencies_follow_the_probabilities():
    """2000 draws at equidistant u: the token histogram is the warped distribution (a property of the inverse CDF, checked end to end)."""
    with emulated_backend():
        from videollama2_amd import ops
        V, n = 300, 400
        x = _logits(V, 9, spread=2.0)
        pr = SO.probs(x, 0.8, 20, 0.9)
        us = ((torch.arange(n).float() + 0.5) / n)
        tok = torch.zeros(1, dtype=torch.int32)
        hist = torch.zeros(n, dtype=torch.int32)
        for i in range(n):
            ops.sample_token(x, tok, us, 0.8, 20, 0.9, hist=hist, step=i)
        freq = torch.bincount(hist.long(), minlength=V).float() / n
        assert (freq - pr).abs().max() < 1.5 / n + 1e-4


def test_emu_generate_with_do_sample_follows_the_oracle_step_by_step():
    """model.generate(do_sample=True, temperature, top_k, top_p) end to end on the small golden config (emulated kernels): with the SAME uniform
    numbers, every sampled token is the oracle's draw from the oracle's warped distribution of OUR logits of that step (teacher-forced by construction:
    the product feeds its own token back), the run is repeatable under a seeded generator, and top_k = 1 reproduces the greedy stream."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = torch.load(os.path.join(root, "tests", "golden", "small_T4.pt"), weights_only=False)
    from oracle import vl2_oracle as O
    with emulated_backend():
        from videollama2_amd.model import VideoLLaMA2Hip
        cfg = g["cfg"]
        m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"], round_bf16=True), "cpu", max_seq_len=64)
        ids = g["input_ids"][None]
        kw = dict(images=[(g["frames"], "video")], attention_mask=torch.ones_like(ids), max_new_tokens=5, return_logits=True)
        greedy, _ = m.generate(ids, do_sample=False, **kw)
        k1, _ = m.generate(ids, do_sample=True, temperature=0.7, top_k=1, top_p=1.0, **kw)
        assert k1.tolist() == greedy.tolist()
        T, k, p = 1.5, 40, 0.9
        gen = torch.Generator().manual_seed(11)
        out, logits = m.generate(ids, do_sample=True, temperature=T, top_k=k, top_p=p, generator=gen, **kw)
        us = torch.rand((5,), generator=torch.Generator().manual_seed(11))
        for s in range(out.shape[1]):
            pr = SO.probs(logits[s], T, k, p)
            t = int(out[0, s])
            assert pr[t] > 0 or t in SO.boundary_tokens(logits[s], T, k, p)
            cdf = pr.double().cumsum(0)
            lo, hi = (float(cdf[t - 1]) if t > 0 else 0.0), float(cdf[t])
            assert lo - 2e-5 <= float(us[s]) * float(cdf[-1]) <= hi + 2e-5, (s, t, float(us[s]), lo, hi)
        again, _ = m.generate(ids, do_sample=True, temperature=T, top_k=k, top_p=p, generator=torch.Generator().manual_seed(11), **kw)
        assert again.tolist() == out.tolist()
        with pytest.raises(ValueError, match="temperature"):
            m.generate(ids, do_sample=True, temperature=0.0, **kw)
