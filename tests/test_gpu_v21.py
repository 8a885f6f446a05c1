"""VideoLLaMA2.1 family on MI355X: SigLIP-so400m tower + stc_connector_v35 + Qwen2-7B decoder.

(1) Golden fixture minted from the REAL reference (tests/golden/small_v21_T4.pt): tower -> connector -> splice -> prefill ->
    greedy tokens.  (2) Full-width slices against the fp32 oracle: one SigLIP layer at hidden 1152 / 16 heads x 72 / MLP 4304 /
    384^2 frames (exercises the head and MLP zero-padding at the real sizes), the v35 connector at 27x27 -> 13x13, two Qwen2-7B
    layers (28 q heads over 4 kv heads: decode attention in two head blocks; q/k/v bias) prefill + decode.
Tolerances: as tests/test_gpu_stages.py (no worse than twice the reference's own bf16 floor)."""
import pytest
import torch

from oracle import vl2_oracle as O
from tests.test_gpu_stages import FULL_TOL, stage_ok
from tests.util import rel, sd_to, token_tie_ok

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def small21(golden_small_v21):
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small_v21
    cfg = g["cfg"]
    sd = O.seeded_state_dict(cfg, g["seed"], round_bf16=True)
    return g, cfg, sd, VideoLLaMA2Hip(cfg, sd, DEV, max_seq_len=256)


def test_small_golden_v21_tower_connector_generate(small21):
    g, cfg, sd, model = small21
    rec = []
    frames = g["frames"]
    sd16 = sd_to(sd, torch.bfloat16)
    with torch.no_grad():
        t16, hs16 = O.siglip_tower(sd16, cfg, frames.bfloat16(), True)
        f16, st16 = O.stc_connector(sd16, t16.view(1, *t16.shape), return_stages=True, padding=0)
    tower = model.vision_tower(frames.to(DEV))
    assert tower.dtype == frames.dtype and tuple(tower.shape) == tuple(g["tower_out"].shape)
    stage_ok("siglip tower_out", tower, g["tower_out"], t16, rec)
    out, st = model.mm_projector(tower.view(1, *tower.shape), return_stages=True)
    stage_ok("v35 stc s1", st["s1"].permute(0, 3, 1, 2), g["stc_s1"], st16["s1"], rec)
    stage_ok("v35 stc sampler", st["sampler"].permute(3, 0, 1, 2)[None], g["stc_sampler"], st16["sampler"], rec)
    stage_ok("v35 mm_features", out, g["mm_features"], f16, rec)
    assert tuple(out.shape) == (1, O.n_visual_tokens(4, 4, padding=0), cfg["llm"]["hidden_size"])
    ids = g["input_ids"][None].to(DEV)
    for use_graph in (False, True):
        _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, torch.ones_like(ids), None, None, [(frames.to(DEV), "video")])
        toks, logits = model.decoder.generate(emb[0], max_new_tokens=8, return_logits=True, use_graph=use_graph)
        assert rel(logits[0], g["step_logits"][0]) < 2.5e-2
        ref = g["new_tokens"].tolist()
        got = toks[0].tolist()
        for i, (a, b) in enumerate(zip(got, ref)):                    # tokens must agree while the fp32 top-2 margin is clear
            if a != b:
                ok, margin, dmax = token_tie_ok(logits[i], g["step_logits"][i])
                assert ok, (use_graph, i, got, ref, margin, dmax)
                break


def test_full_width_siglip_layer():
    """SigLIP-so400m widths, T=2 frames of 384^2, ONE encoder layer (select_layer=-2 on a 2-layer tower)."""
    from videollama2_amd.tower import HipSiglipVisionTower
    cfg = O.config_videollama2_1_7b_16f(2)
    cfg["vision"]["num_hidden_layers"] = 2
    sd = O.seeded_state_dict(cfg, 7, only=lambda n: "vision_tower" in n and ".head." not in n)
    frames = O.normalise_frames_u8_siglip(torch.randint(0, 256, (2, 384, 384, 3), dtype=torch.uint8,
                                                        generator=torch.Generator().manual_seed(0)).numpy())
    with torch.no_grad():
        ref = O.siglip_tower(sd, cfg, frames.bfloat16().float())
    tower = HipSiglipVisionTower(cfg, sd, DEV)
    assert tower.w["hd"] == 72 and tower.w["hdp"] == 96 and tower.w["layers"][0]["w1"].shape == (4352, 1152)
    out = tower(frames.to(DEV))
    assert tuple(out.shape) == (2, 729, 1152) and tower.num_patches == 729
    stage_ok("full-width SigLIP layer", out, ref, FULL_TOL["vit_layer"], [])


def test_full_width_v35_connector():
    """stc_connector_v35 at VideoLLaMA2.1-7B widths (1152 -> 3584) on T=4 frames of 27x27 tokens -> (2,13,13) = 338 tokens."""
    from videollama2_amd.connector import HipSTCConnector
    cfg = O.config_videollama2_1_7b_16f(4)
    sd = O.seeded_state_dict(cfg, 11, only=lambda n: "mm_projector" in n)
    x = (torch.randn(1, 4, 729, 1152, generator=torch.Generator().manual_seed(1))).bfloat16().float()
    with torch.no_grad():
        ref, st = O.stc_connector(sd, x, return_stages=True, padding=0)
    conn = HipSTCConnector(sd, DEV, padding=0)
    out, mine = conn(x.to(DEV), return_stages=True)
    rec = []
    stage_ok("full-width v35 s1", mine["s1"].permute(0, 3, 1, 2), st["s1"], FULL_TOL["stc"], rec)
    stage_ok("full-width v35 sampler", mine["sampler"].permute(3, 0, 1, 2)[None], st["sampler"], FULL_TOL["stc"], rec)
    stage_ok("full-width v35 out", out, ref, FULL_TOL["stc"], rec)
    assert tuple(out.shape) == (1, O.n_visual_tokens(4, 27, padding=0), 3584) == (1, 338, 3584)


def test_full_width_qwen2_layers_prefill_and_decode():
    """Qwen2-7B widths (3584, 28 q / 4 kv heads, MLP 18944, q/k/v bias), 2 layers, vocab cut to 4096 for the CPU oracle,
    S=300 prefill + 3 decode steps, logits vs the fp32 oracle."""
    from videollama2_amd.decoder import HipQwen2Decoder
    cfg = O.config_videollama2_1_7b_16f(16)
    cfg["llm"]["num_hidden_layers"] = 2
    cfg["llm"]["vocab_size"] = 4096
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    assert "model.layers.0.self_attn.q_proj.bias" in sd
    S = 300
    x = torch.randn(S, 3584, generator=torch.Generator().manual_seed(2)).bfloat16().float() * 0.5
    with torch.no_grad():
        toks, lg = O.greedy_generate(sd, cfg, x, 4)
    dec = HipQwen2Decoder(cfg, sd, DEV, max_seq_len=512)
    out, mine = dec.generate(x.to(DEV), max_new_tokens=4, return_logits=True)
    rec = []
    stage_ok("qwen2 full-width prefill logits", mine[0], lg[0], FULL_TOL["logits"], rec)
    if out[0].tolist() == toks:
        for s in range(1, 4):
            stage_ok(f"qwen2 full-width decode logits {s}", mine[s], lg[s], FULL_TOL["logits"], rec)
    else:
        print("[parity] greedy path diverged on a near-tie:", out[0].tolist(), toks)
        s_div = next(i for i, (a, b) in enumerate(zip(out[0].tolist(), toks)) if a != b)
        ok, margin, dmax = token_tie_ok(mine[s_div], lg[s_div])
        assert ok, f"step {s_div}: token differs although fp32 top-2 margin {margin:.3e} >= 2 x max|dlogit| {dmax:.3e}"
    graph = dec.generate(x.to(DEV), max_new_tokens=4, use_graph=True)
    assert graph.tolist() == out.tolist()


def test_72b_width_decoder_layer_single_and_tp_shard():
    """Qwen2-72B widths (hidden 8192, 64 q / 8 kv heads, MLP 29568, q/k/v bias), ONE layer, vocab cut to 2048 for the CPU
    oracle: S=96 prefill + 2 decode steps vs the fp32 oracle; and the TP=8 shard of rank 3 (MLP slice 3696 zero-padded to
    3712) loads and runs (its output is a partial sum, only shapes are checked)."""
    from videollama2_amd.config import videollama2_72b
    from videollama2_amd.decoder import HipQwen2Decoder
    cfg = videollama2_72b(16)
    cfg["llm"]["num_hidden_layers"] = 1
    cfg["llm"]["vocab_size"] = 2048
    ocfg = dict(vision=O.config_videollama2_7b(16)["vision"], llm=dict(cfg["llm"]), num_frames=16)
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(ocfg, 9, only=keep)
    S = 96
    x = torch.randn(S, 8192, generator=torch.Generator().manual_seed(2)).bfloat16().float() * 0.5
    with torch.no_grad():
        toks, lg = O.greedy_generate(sd, ocfg, x, 3)
    dec = HipQwen2Decoder(cfg, sd, DEV, max_seq_len=256)
    out, mine = dec.generate(x.to(DEV), max_new_tokens=3, return_logits=True)
    rec = []
    stage_ok("72B-width prefill logits", mine[0], lg[0], FULL_TOL["logits"], rec)
    if out[0].tolist() == toks:
        for s in range(1, 3):
            stage_ok(f"72B-width decode logits {s}", mine[s], lg[s], FULL_TOL["logits"], rec)
    shard = HipQwen2Decoder(cfg, sd, DEV, max_seq_len=256, tp_shard=(3, 8))
    assert shard.nh == 8 and shard.nkv == 1 and shard.w["layers"][0]["wd"].shape == (8192, 3712)
    assert shard.w["layers"][0]["wgu"].shape == (2 * 3712, 8192)
    part = shard.prefill(x.to(DEV))
    assert part.shape == (2048,) and torch.isfinite(part).all()


def test_batched_decode_equals_sequential_full_width():
    """Batched decode at Mistral-7B widths (2 layers): 4 requests of different lengths decoded together vs one at a time --
    same tokens, bit-identical logits (the multi-row GEMV streams the weights once for the 4 tokens of a step)."""
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    g = torch.Generator().manual_seed(8)
    embeds = [(torch.randn(n, 4096, generator=g) * 0.5).bfloat16().to(DEV) for n in (300, 77, 129, 1)]
    seq = [dec.generate(e, max_new_tokens=6, return_logits=True) for e in embeds]
    outs, blogits = dec.generate_batch(embeds, max_new_tokens=6, return_logits=True)
    for b, (toks, logits) in enumerate(seq):
        assert outs[b].tolist() == toks[0].tolist(), b
        assert torch.equal(blogits[:, b], logits), b
    again = dec.generate_batch(embeds[:3], max_new_tokens=6)                 # smaller batch on the same buffers
    assert [o.tolist() for o in again] == [s[0][0].tolist() for s in seq[:3]]
    eager, elogits = dec.generate_batch(embeds, max_new_tokens=6, return_logits=True, use_graph=False)
    assert [o.tolist() for o in eager] == [o.tolist() for o in outs] and torch.equal(elogits, blogits)   # hipGraph == eager


def test_batched_decode_gemm_path_matches_sequential_to_rounding():
    """8 requests: the batched step runs its projections as MFMA GEMMs with M = 8 (weights streamed once for all of them).
    Same arithmetic as a prefill row, so the logits equal the single-sequence decode to bf16 rounding and the greedy tokens
    agree wherever the top-2 margin is clear."""
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    g = torch.Generator().manual_seed(9)
    embeds = [(torch.randn(n, 4096, generator=g) * 0.5).bfloat16().to(DEV) for n in (200, 31, 77, 5, 129, 64, 1, 300)]
    assert len(embeds) >= dec.GEMM_BATCH
    seq = [dec.generate(e, max_new_tokens=4, return_logits=True) for e in embeds]
    outs, blogits = dec.generate_batch(embeds, max_new_tokens=4, return_logits=True)
    for b, (toks, logits) in enumerate(seq):
        assert torch.equal(blogits[0, b], logits[0])                          # step 0 is the prefill: identical
        for s in range(1, blogits.shape[0]):
            if outs[b][:s].tolist() != toks[0, :s].tolist():
                break                                                         # diverged on a near-tie earlier: later steps differ
            assert rel(blogits[s, b], logits[s]) < 2e-2, (b, s)
            if outs[b][s].item() != toks[0, s].item():                        # batched (skinny-MFMA) vs single-request arithmetic
                ok, margin, dmax = token_tie_ok(blogits[s, b], logits[s])
                assert ok, (b, s, margin, dmax)


def test_full_width_72b_connector():
    """stc_connector at the 72B decoder's width (1024 -> 8192 channels, 1.93 B parameters) on T=2 frames vs the fp32 oracle:
    the 8192-channel LayerNorm / depthwise / SE kernels and the K = 65536 gathered Conv3d GEMM in place."""
    from videollama2_amd.config import videollama2_72b
    from videollama2_amd.connector import HipSTCConnector
    cfg = videollama2_72b(2)
    ocfg = dict(vision=O.config_videollama2_7b(2)["vision"], llm=dict(cfg["llm"]), num_frames=2)
    sd = O.seeded_state_dict(ocfg, 13, only=lambda n: "mm_projector" in n)
    x = (torch.randn(1, 2, 576, 1024, generator=torch.Generator().manual_seed(1))).bfloat16().float()
    with torch.no_grad():
        ref, st = O.stc_connector(sd, x, return_stages=True)
    conn = HipSTCConnector(sd, DEV)
    out, mine = conn(x.to(DEV), return_stages=True)
    rec = []
    stage_ok("72B-width stc s1", mine["s1"].permute(0, 3, 1, 2), st["s1"], FULL_TOL["stc"], rec)
    stage_ok("72B-width stc sampler", mine["sampler"].permute(3, 0, 1, 2)[None], st["sampler"], FULL_TOL["stc"], rec)
    stage_ok("72B-width stc out", out, ref, FULL_TOL["stc"], rec)
    assert tuple(out.shape) == (1, 2 * 169, 8192)


def test_generate_batch_with_videos_equals_one_by_one(small21):
    """Model-level batch: three video requests (one tower call for all frames, one prefill pass over the concatenated prompts,
    batched decode) + a text-only request give each request exactly the tokens it gets alone."""
    g, cfg, sd, model = small21
    V = cfg["llm"]["vocab_size"]
    rng = torch.Generator().manual_seed(5)
    reqs = []
    for n_pre, n_post, shift in ((7, 9, 0), (3, 2, 1), (1, 12, 2)):
        ids = torch.cat([torch.tensor([1]), torch.randint(3, V, (n_pre,), generator=rng), torch.tensor([-201]),
                         torch.randint(3, V, (n_post,), generator=rng)])
        reqs.append((ids[None].to(DEV), [(torch.roll(g["frames"], shift, dims=0).to(DEV), "video")]))
    reqs.append((torch.tensor([[1, 17, 99, 5]], device=DEV), None))
    alone = [model.generate(ids, images=images, do_sample=False, max_new_tokens=6, attention_mask=torch.ones_like(ids))[0].tolist()
             for ids, images in reqs]
    together = [o.tolist() for o in model.generate_batch(reqs, max_new_tokens=6)]
    assert together == alone
