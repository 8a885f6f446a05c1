"""Stage-level and end-to-end parity on MI355X.

(1) Golden fixtures minted from the REAL reference (tests/golden/small_T4.pt, oracle/make_golden.py): frames ->
    ViT hidden states -> STC stages -> visual tokens -> spliced embeddings -> prefill logits -> greedy tokens.
(2) Full-width slices (VideoLLaMA2-7B dims) against the fp32 oracle computed on the host in-test.

Contract: bf16 storage between kernels means the end-to-end error cannot reach 1e-3; every stage
must be no worse than the reference's OWN bf16 path on the same inputs (oracle run in bf16 = the noise floor,
measured here) -- asserted as err <= max(2 * floor, 4e-3) -- and greedy tokens must match wherever the fp32 top-2
margin exceeds the logit error."""
import pytest
import torch

from oracle import vl2_oracle as O
from tests.util import rel, sd_to, token_tie_ok

pytestmark = pytest.mark.gpu
DEV = "cuda"


# Full-width slices: running the oracle in bf16 on an unknown host CPU can take minutes, so the floors measured in
# SURVEY.md 7.3-6 on the reference's own modules are used instead (ViT 4.4e-3 after layer 1, STC 1.7e-2, logits ~1.8e-2).
FULL_TOL = dict(vit_layer=6e-3, stc=1.7e-2, logits=2e-2)


def stage_ok(name, ours, ref32, ref16, record):
    e = rel(ours, ref32)
    floor = rel(ref16, ref32) if torch.is_tensor(ref16) else float(ref16) / 2.0
    record.append((name, e, floor))
    print(f"[parity] {name:28s} ours {e:.3e}   reference-bf16 floor {floor:.3e}")
    assert e <= max(2.0 * floor, 4e-3), f"{name}: rel-L2 {e:.3e} vs bf16 floor {floor:.3e}"


@pytest.fixture(scope="module")
def small(golden_small):
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    sd = O.seeded_state_dict(cfg, g["seed"], round_bf16=True)
    model = VideoLLaMA2Hip(cfg, sd, DEV, max_seq_len=256)
    return g, cfg, sd, model


def test_small_golden_tower_and_connector(small):
    g, cfg, sd, model = small
    rec = []
    frames = g["frames"]
    sd16 = sd_to(sd, torch.bfloat16)
    with torch.no_grad():
        t16 = O.clip_tower(sd16, cfg, frames.bfloat16())
        f16, st16 = O.stc_connector(sd16, t16.view(1, *t16.shape), return_stages=True)
    x, T, N1 = model.vision_tower.forward_hidden(frames.to(DEV))
    stage_ok("vit hidden[-2] (with CLS)", x.view(T, N1, -1), g["vit_hidden"][-1], O.clip_tower(sd16, cfg, frames.bfloat16(), True)[1][-1], rec)
    tower = model.vision_tower(frames.to(DEV))
    assert tower.dtype == frames.dtype and tuple(tower.shape) == tuple(g["tower_out"].shape)
    stage_ok("tower_out", tower, g["tower_out"], t16, rec)
    out, st = model.mm_projector(tower.view(1, *tower.shape), return_stages=True)
    stage_ok("stc s1", st["s1"].permute(0, 3, 1, 2), g["stc_s1"], st16["s1"], rec)
    stage_ok("stc sampler", st["sampler"].permute(3, 0, 1, 2)[None], g["stc_sampler"], st16["sampler"], rec)
    stage_ok("stc s2", st["s2"].permute(0, 3, 1, 2), g["stc_s2"], st16["s2"], rec)
    stage_ok("mm_features", out, g["mm_features"], f16, rec)
    assert tuple(out.shape) == (1, O.n_visual_tokens(cfg["num_frames"], 4), cfg["llm"]["hidden_size"])


def test_small_golden_splice_prefill_generate(small):
    g, cfg, sd, model = small
    rec = []
    ids = g["input_ids"][None]
    _, mask, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, torch.ones_like(ids), None, None,
                                                                      [(g["frames"].to(DEV), "video")])
    assert tuple(emb.shape[1:]) == tuple(g["inputs_embeds"].shape) and tuple(mask.shape) == (1, emb.shape[1])
    sd16 = sd_to(sd, torch.bfloat16)
    with torch.no_grad():
        f16 = O.encode_images_or_videos(sd16, cfg, [(g["frames"].bfloat16(), "video")])
        e16 = O.splice_inputs_embeds(sd16, g["input_ids"], [f16[0]])
        l16, _ = O.mistral_forward(sd16, cfg, e16, last_only=False)
    stage_ok("inputs_embeds", emb[0], g["inputs_embeds"], e16, rec)
    # text rows are exact gathers of bf16 weights
    n_text_head = 8
    assert torch.equal(emb[0, :n_text_head].float().cpu(), g["inputs_embeds"][:n_text_head])
    logits = model.decoder.prefill(emb[0], return_all_logits=True)
    stage_ok("prefill logits", logits, g["prefill_logits"], l16, rec)
    # greedy decode, teacher-free; compare with the reference's tokens where the fp32 margin is decisive
    out, step_logits = model.generate(ids, images=[(g["frames"].to(DEV), "video")], do_sample=False, max_new_tokens=8,
                                      attention_mask=torch.ones_like(ids), return_logits=True)
    gold_tok, gold_logits = g["new_tokens"].tolist(), g["step_logits"]
    ours = out[0].tolist()
    for s, (a, b) in enumerate(zip(ours, gold_tok)):
        top2 = gold_logits[s].topk(2).values
        margin = (top2[0] - top2[1]).item()
        err = (step_logits[s].float().cpu() - gold_logits[s]).abs().max().item()
        print(f"[parity] step {s}: ours {a} ref {b} margin {margin:.3e} max|dlogit| {err:.3e}")
        if a != b:
            assert margin < 2 * err, f"step {s}: token {a} != {b} although margin {margin:.3e} > 2*err {err:.3e}"
            break
    assert rel(step_logits[0], gold_logits[0]) < 2e-2


def test_full_width_vit_layer_and_embeddings():
    """VideoLLaMA2-7B widths, T=2 frames, ONE encoder layer (select_layer=1 on a 1-layer tower)."""
    from videollama2_amd.tower import HipCLIPVisionTower
    cfg = O.config_videollama2_7b(2)
    cfg["vision"]["num_hidden_layers"] = 2
    cfg["vision"]["select_layer"] = -2            # -> runs layer 0 only
    sd = O.seeded_state_dict(cfg, 7, only=lambda n: "vision_tower" in n)
    frames = O.normalise_frames_u8(torch.randint(0, 256, (2, 336, 336, 3), dtype=torch.uint8,
                                                 generator=torch.Generator().manual_seed(0)).numpy())
    with torch.no_grad():
        ref, hs = O.clip_tower(sd, cfg, frames.bfloat16().float(), return_hidden=True)
    r16 = FULL_TOL["vit_layer"]
    tower = HipCLIPVisionTower(cfg, sd, DEV)
    out = tower(frames.to(DEV))
    rec = []
    stage_ok("full-width ViT layer", out, ref, r16, rec)
    assert tuple(out.shape) == (2, 576, 1024)


def test_full_width_stc_connector():
    """Full STC connector (hidden 4096, 489 M params) on T=2 frames: 24x24 -> (2,13,13) = 338 tokens."""
    from videollama2_amd.connector import HipSTCConnector
    cfg = O.config_videollama2_7b(2)
    sd = O.seeded_state_dict(cfg, 11, only=lambda n: "mm_projector" in n)
    x = (torch.randn(1, 2, 576, 1024, generator=torch.Generator().manual_seed(1))).bfloat16().float()
    with torch.no_grad():
        ref, st = O.stc_connector(sd, x, return_stages=True)
    r16 = FULL_TOL["stc"]
    st16 = dict(s1=r16, sampler=r16)
    conn = HipSTCConnector(sd, DEV)
    out, mine = conn(x.to(DEV), return_stages=True)
    rec = []
    stage_ok("full-width stc s1", mine["s1"].permute(0, 3, 1, 2), st["s1"], st16["s1"], rec)
    stage_ok("full-width stc sampler", mine["sampler"].permute(3, 0, 1, 2)[None], st["sampler"], st16["sampler"], rec)
    stage_ok("full-width stc out", out, ref, r16, rec)
    assert tuple(out.shape) == (1, 2 * 169, 4096)


def test_full_width_mistral_layers_prefill_and_decode():
    """Mistral-7B widths, 2 layers, S=300 prefill + 3 decode steps, logits vs the fp32 oracle."""
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    S = 300
    x = torch.randn(S, 4096, generator=torch.Generator().manual_seed(2)).bfloat16().float() * 0.5
    with torch.no_grad():
        toks, lg = O.greedy_generate(sd, cfg, x, 4)
    lg16 = [FULL_TOL["logits"]] * 4
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    out, mine = dec.generate(x.to(DEV), max_new_tokens=4, return_logits=True)
    rec = []
    stage_ok("full-width prefill logits", mine[0], lg[0], lg16[0], rec)
    if out[0].tolist() == toks:
        for s in range(1, 4):
            stage_ok(f"full-width decode logits {s}", mine[s], lg[s], lg16[s], rec)
    else:
        print("[parity] greedy path diverged on a near-tie:", out[0].tolist(), toks)
        s_div = next(i for i, (a, b) in enumerate(zip(out[0].tolist(), toks)) if a != b)       # first step that differs
        ok, margin, dmax = token_tie_ok(mine[s_div], lg[s_div])
        assert ok, f"step {s_div}: token differs although fp32 top-2 margin {margin:.3e} >= 2 x max|dlogit| {dmax:.3e}"


def test_hipgraph_decode_matches_eager(small):
    """The captured {argmax + decode step} hipGraph must reproduce the eager loop token for token and logit for logit."""
    g, cfg, sd, model = small
    emb = g["inputs_embeds"].to(DEV)
    eager, le = model.decoder.generate(emb, max_new_tokens=8, return_logits=True)
    graph, lg = model.decoder.generate(emb, max_new_tokens=8, return_logits=True, use_graph=True)
    assert eager.tolist() == graph.tolist()
    assert torch.equal(le, lg)
    graph2 = model.decoder.generate(emb, max_new_tokens=5, use_graph=True)       # replay the same graph on a fresh request
    assert graph2[0].tolist() == eager[0, :5].tolist()


@pytest.mark.parametrize("T", [8, 16, 32])
def test_token_count_law_and_connector_parity_over_T(T):
    """N_vis = (T/2+1) * (g/2+1)^2 for T = 8/16/32 (845/1521/2873 at g = 24) through the HIP connector, small
    widths / 4x4 grid so the oracle runs in a second; also an image (modal 'image' -> expanded to num_frames)."""
    from videollama2_amd.model import VideoLLaMA2Hip
    cfg = O.config_small(T)
    sd = O.seeded_state_dict(cfg, 3)
    model = VideoLLaMA2Hip(cfg, sd, DEV, max_seq_len=256)
    frames = torch.randn(T, 3, 56, 56, generator=torch.Generator().manual_seed(T)).bfloat16().float()
    with torch.no_grad():
        ref = O.encode_images_or_videos(sd, cfg, [(frames, "video")])
        ref_img = O.encode_images_or_videos(sd, cfg, [(frames[:1], "image")])
    out = model.encode_images_or_videos([(frames.to(DEV), "video")])
    assert tuple(out.shape) == tuple(ref.shape) == (1, (T // 2 + 1) * 9, cfg["llm"]["hidden_size"])
    assert rel(out, ref) < 3e-2
    img = model.encode_images_or_videos([(frames[:1].to(DEV), "image")])
    assert tuple(img.shape) == tuple(ref_img.shape) and rel(img, ref_img) < 3e-2
    assert [O.n_visual_tokens(t) for t in (8, 16, 32)] == [845, 1521, 2873]


def test_vit_frame_chunks_on_streams_match_single_stream():
    """tower.streams > 1 runs the frames as interleaved chunks on separate HIP streams: same kernels on row subsets
    (every GEMM kernel accumulates K in the same order), so the result must be identical, not merely close."""
    from videollama2_amd.tower import HipCLIPVisionTower
    cfg = O.config_videollama2_7b(12)
    cfg["vision"]["num_hidden_layers"] = 3
    sd = O.seeded_state_dict(cfg, 11, only=lambda n: "vision_tower" in n)
    tower = HipCLIPVisionTower(cfg, sd, DEV)
    frames = torch.randn(12, 3, 336, 336, generator=torch.Generator().manual_seed(3)).bfloat16().to(DEV)
    tower.streams = 1
    one = tower(frames)
    for ns in (2, 3):
        tower.streams = ns
        for _ in range(2):                       # second pass reuses the side streams and cached buffers
            many = tower(frames)
            torch.cuda.synchronize()
            assert torch.equal(one, many), f"{ns} streams: max |d| = {(one.float() - many.float()).abs().max().item():.3e}"


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_connector_cut_rank_by_rank_equals_unsharded(world):
    """Full-width STC + a 2-layer tower, T=8: the rank-local pieces of dist.py's sharded-connector cut, run for every rank
    in this process (halo handed over by reference), must reproduce the unsharded encoder (bit for bit while no GEMM splits K)."""
    from videollama2_amd.connector import HipSTCConnector
    from videollama2_amd.dist import FrameSharder
    from videollama2_amd.tower import HipCLIPVisionTower
    cfg = O.config_videollama2_7b(8)
    cfg["vision"]["num_hidden_layers"] = 3
    sd = O.seeded_state_dict(cfg, 5, only=lambda n: "vision_tower" in n or "mm_projector" in n)
    tower, conn = HipCLIPVisionTower(cfg, sd, DEV), HipSTCConnector(sd, DEV)
    frames = torch.randn(8, 3, 336, 336, generator=torch.Generator().manual_seed(4)).bfloat16().to(DEV)
    from videollama2_amd import ops
    ref = conn(tower(frames).view(1, 8, 576, 1024))      # same kernels on row subsets -> identical
    out = FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world)
    assert out.shape == ref.shape == (1, (8 // 2 + 1) * 13 * 13, 4096)
    assert torch.equal(out, ref), f"max |d| = {(out.float() - ref.float()).abs().max().item():.3e}"
    try:                                   # opt-in split-K reorders fp32 partial sums: two bf16 pipelines that round differently
        ops.set_splitk(True)               # end up one bf16 noise floor apart (the STC floor is 1.7e-2)
        out2 = FrameSharder.encode_video_all_ranks_locally(tower, conn, frames, world)
    finally:
        ops.set_splitk(False)
    assert rel(out2, ref) < 3.4e-2, f"with split-K: rel {rel(out2, ref):.3e}"
