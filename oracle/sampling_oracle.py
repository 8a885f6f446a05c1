"""TEST INFRASTRUCTURE ONLY (never imported by videollama2_amd/): the sampled decode step of the reference, restated.

`mm_infer(..., do_sample=True, temperature=, top_p=)` (/root/reference/videollama2/__init__.py:93-106) hands those arguments to HF
`GenerationMixin.generate`, which builds the logits-warper list of transformers/generation/utils.py `_get_logits_processor` -- in this order:
TemperatureLogitsWarper (temperature != 1), TopKLogitsWarper (generation_config.top_k, 50 unless the checkpoint says otherwise),
TopPLogitsWarper (top_p < 1) -- and then, in `_sample`, draws `torch.multinomial(softmax(scores), 1)`.

Restated here on one row of fp32 logits (transformers/generation/logits_process.py, pinned 4.40.0 in the reference's requirements; the
installed version's classes are the pin of tests/test_sampling.py):
    warp(logits, temperature, top_k, top_p) -> scores with -inf on the removed tokens
    probs(...)                               -> softmax of the warped scores (what multinomial draws from)
    draw(probs, u)                           -> the token whose interval of the cumulative distribution, in token-index order, holds u
The draw is OUR definition: torch.multinomial's consumption of its Philox stream is not reproducible outside torch (and differs between CPU and
GPU builds), so the parity statement is about the kept set and the probabilities; a token is then a deterministic function of (probs, u)."""
import torch


def warp(logits, temperature=1.0, top_k=50, top_p=1.0, min_tokens_to_keep=1):
    s = logits.detach().float().cpu().clone()
    if temperature != 1.0:
        s = s / temperature                                                      # TemperatureLogitsWarper.__call__
    V = s.numel()
    if top_k and top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), V)
        kth = torch.topk(s, k)[0][-1]                                            # TopKLogitsWarper.__call__
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p < 1.0:
        sl, si = torch.sort(s, descending=False)                                 # TopPLogitsWarper.__call__
        cum = sl.softmax(dim=-1).cumsum(dim=-1)
        rem = cum <= (1 - top_p)
        rem[-min_tokens_to_keep:] = False
        mask = torch.zeros(V, dtype=torch.bool).scatter(0, si, rem)
        s = s.masked_fill(mask, float("-inf"))
    return s


def probs(logits, temperature=1.0, top_k=50, top_p=1.0):
    return torch.softmax(warp(logits, temperature, top_k, top_p), dim=-1)


def draw(p, u):
    """Inverse CDF in token-index order (float64 cumulative sums): the first token whose cumulative probability exceeds u * total."""
    c = p.double().cumsum(0)
    t = float(u) * float(c[-1])
    i = int(torch.searchsorted(c, torch.tensor(t, dtype=torch.float64), right=True))
    return min(i, p.numel() - 1)


def boundary_tokens(logits, temperature, top_k, top_p, tol=2e-6):
    """Tokens whose membership of the top-p set hangs on less than `tol` of cumulative probability (an fp32 cumsum in another order may flip them):
    a comparison of kept sets ignores these."""
    s = logits.detach().float().cpu() / temperature
    if top_k and 0 < top_k < s.numel():
        s = s.masked_fill(s < torch.topk(s, top_k)[0][-1], float("-inf"))
    if top_p >= 1.0:
        return set()
    sl, si = torch.sort(s, descending=False)
    cum = sl.double().softmax(dim=-1).cumsum(dim=-1)
    near = (cum - (1 - top_p)).abs() < tol
    return set(si[near].tolist())
