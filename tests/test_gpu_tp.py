"""Tensor-parallel decoder with >= 2 REAL shards on ONE MI355X (the reference's counterpart is accelerate's
`device_map="auto"`, videollama2/model/__init__.py:54): every rank's shard of two full-width Mistral-7B layers is built in this
process, runs its own prefill / decode kernels on the GPU, and the row-parallel partial sums (o_proj, down_proj) are summed where
the RCCL all-reduce would sit (`dist.LocalTensorParallel`: a rendezvous of one host thread per rank, rank-ordered fp32 sum).
Checked: every rank ends with the same logits bits; the logits match the fp32 oracle and the unsharded decoder to the bf16 floor;
greedy tokens agree wherever the fp32 top-2 margin exceeds twice the logit error.  The collectives themselves (RCCL over xGMI) are
exercised at world 1 in tests/test_gpu_rccl.py and across 2 processes on CPU in tests/test_dist_gloo.py."""
import pytest
import torch

from oracle import vl2_oracle as O
from tests.util import rel, token_tie_ok

DEV = "cuda"


def bf16_floor(cfg, sd, x, toks, lg, dev):
    """rel-L2 of the REFERENCE's own bf16 arithmetic (the oracle restatement run in bf16 on torch-ROCm) against the fp32 oracle, per step
    (prefill + teacher-forced decode): the floor the sharded decoder is judged against, measured on the same weights and inputs."""
    sd16 = {k: v.to(device=dev, dtype=torch.bfloat16) for k, v in sd.items()}
    out = []
    with torch.no_grad():
        l16, caches = O.mistral_forward(sd16, cfg, x.to(dev).bfloat16(), 0, None)
        out.append(rel(l16[0].float(), lg[0]))
        for s in range(len(toks) - 1):
            xt = torch.nn.functional.embedding(torch.tensor([toks[s]], device=dev), sd16["model.embed_tokens.weight"])
            l16, caches = O.mistral_forward(sd16, cfg, xt, x.shape[0] + s, caches)
            out.append(rel(l16[0].float(), lg[s + 1]))
    return out


def run_local_tp(cfg, sd, x, toks, lg, R, max_seq_len, dev):
    from videollama2_amd.decoder import HipMistralDecoder
    from videollama2_amd.dist import LocalTensorParallel
    n_dec = len(toks) - 1
    floor = bf16_floor(cfg, sd, x, toks, lg, dev)
    one = HipMistralDecoder(cfg, sd, dev, max_seq_len=max_seq_len)
    ref = [one.prefill(x.to(dev)).clone()]
    for s in range(n_dec):
        one.tok.copy_(torch.tensor([toks[s]], dtype=torch.int32))
        ref.append(one.decode_step().clone())
    del one
    ltp = LocalTensorParallel(R)
    shards = [HipMistralDecoder(cfg, sd, dev, max_seq_len=max_seq_len, tp_shard=(r, R), tp_local=ltp) for r in range(R)]
    l = cfg["llm"]
    assert shards[0].nh == l["num_attention_heads"] // R and shards[0].nkv == l["num_key_value_heads"] // R
    assert shards[0].w["layers"][0]["wo"].shape[1] == l["num_attention_heads"] // R * l["head_dim"]

    def program(r):
        def go():
            d = shards[r]
            out = [d.prefill(x.to(dev)).clone()]
            for s in range(n_dec):                                     # teacher-forced on the oracle's tokens
                d.tok.copy_(torch.tensor([toks[s]], dtype=torch.int32))
                out.append(d.decode_step().clone())
            return out
        return go

    outs = ltp.run([program(r) for r in range(R)])
    n_layers = l["num_hidden_layers"]
    assert ltp.reductions == 2 * n_layers * (n_dec + 1), ltp.reductions       # two all-reduces per layer, prefill + every decode step
    rows = []
    for s in range(n_dec + 1):
        for r in range(1, R):
            assert torch.equal(outs[r][s], outs[0][s]), (s, r)                 # replicated lm_head on identical sums: same bits on every rank
        e_tp, e_one, e_rel = rel(outs[0][s], lg[s]), rel(ref[s], lg[s]), rel(outs[0][s], ref[s].float())
        rows.append((e_tp, e_one, e_rel, floor[s]))
        # against the fp32 oracle: the floor rule of the parity suite (twice the reference's own bf16 error on the same inputs, 4e-3 at least);
        # sharded against unsharded: two bf16 paths apart, i.e. within the sum of their distances to fp32
        tol = max(2.0 * floor[s], 4e-3)
        assert e_tp <= tol and e_one <= tol and e_rel <= e_tp + e_one, (R, s, e_tp, e_one, e_rel, floor[s])
        if int(outs[0][s].argmax()) != toks[s]:
            ok, margin, dmax = token_tie_ok(outs[0][s], lg[s])
            assert ok, (R, s, margin, dmax)
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 8])
def test_tp_real_shards_two_full_width_layers(R):
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 41, only=keep)
    x = (torch.randn(300, cfg["llm"]["hidden_size"], generator=torch.Generator().manual_seed(12)) * 0.5).bfloat16().float()
    with torch.no_grad():
        toks, lg = O.greedy_generate(sd, cfg, x, 4)
    rows = run_local_tp(cfg, sd, x, toks, lg, R, 512, DEV)
    print(f"[tp-local] TP={R}: rel-L2 vs fp32 oracle (sharded / unsharded), sharded vs unsharded, reference-bf16 floor per step:",
          [tuple(round(v, 5) for v in r) for r in rows])
