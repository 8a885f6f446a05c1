"""Pins the oracle (oracle/vl2_oracle.py): (a) against golden tensors minted from the REAL reference (travels),
(b) against the reference itself imported in place (only where /root/reference exists)."""
import numpy as np
import pytest
import torch

from oracle import ref_harness as RH
from oracle import vl2_oracle as O


def test_state_dict_names_cover_golden_config(golden_small):
    names = dict(O.state_dict_names(golden_small["cfg"]))
    assert len(names) == 189


def test_oracle_matches_reference_goldens(golden_small):
    g = golden_small
    cfg = g["cfg"]
    sd = O.seeded_state_dict(cfg, g["seed"], round_bf16=True)
    with torch.no_grad():
        assert torch.allclose(O.normalise_frames_u8(g["frames_u8"].numpy()), g["frames"], atol=1e-6)
        out, hs = O.clip_tower(sd, cfg, g["frames"], return_hidden=True)
        for a, b in zip(hs, g["vit_hidden"]):
            assert torch.allclose(a, b, atol=2e-5, rtol=1e-5)
        assert torch.allclose(out, g["tower_out"], atol=2e-5, rtol=1e-5)
        feats, st = O.stc_connector(sd, out.view(1, *out.shape), return_stages=True)
        assert torch.allclose(st["s1"], g["stc_s1"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(st["sampler"], g["stc_sampler"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(st["s2"], g["stc_s2"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(feats, g["mm_features"], atol=2e-5, rtol=1e-5)
        emb = O.splice_inputs_embeds(sd, g["input_ids"], [feats[0]])
        assert torch.allclose(emb, g["inputs_embeds"], atol=2e-5, rtol=1e-5)
        logits, _ = O.mistral_forward(sd, cfg, emb, last_only=False)
        assert torch.allclose(logits, g["prefill_logits"], atol=1e-4, rtol=1e-4)
        toks, step_logits = O.greedy_generate(sd, cfg, emb, 8)
        assert toks == g["new_tokens"].tolist()
        assert torch.allclose(step_logits, g["step_logits"], atol=1e-4, rtol=1e-4)


def test_token_count_law():
    assert [O.n_visual_tokens(t) for t in (8, 16, 32)] == [845, 1521, 2873]


def test_frame_sample_golden(golden_small):
    for (d, n), ids in golden_small["frame_sample"].items():
        assert np.array_equal(O.frame_sample_uniform(d, n), ids.numpy())


@pytest.mark.skipif(not RH.reference_available(), reason="reference tree only exists in the build container")
def test_oracle_matches_live_reference():
    cfg = O.config_small(4)
    model, _ = RH.build_reference_model(cfg)
    RH.reseed_weights(model, 77)
    sd = O.seeded_state_dict(cfg, 77, round_bf16=False)
    ref_sd = {k: v for k, v in model.state_dict().items() if torch.is_floating_point(v)}
    assert set(ref_sd) == set(dict(O.state_dict_names(cfg)))
    rng = np.random.default_rng(5)
    fr = O.normalise_frames_u8(rng.integers(0, 256, (4, 56, 56, 3), dtype=np.uint8))
    with torch.no_grad():
        a = model.encode_images_or_videos([(fr, "video")])
        b = O.encode_images_or_videos(sd, cfg, [(fr, "video")])
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
        ids = torch.tensor([1] + list(rng.integers(3, 512, 5)) + [-201] + list(rng.integers(3, 512, 4)))
        out = model.generate(ids[None], attention_mask=torch.ones_like(ids)[None], images=[(fr, "video")], do_sample=False,
                             max_new_tokens=5, use_cache=True, pad_token_id=0, eos_token_id=None)
        toks, _ = O.generate(sd, cfg, ids, fr, 5)
        assert out[0].tolist() == toks


# ---------------------------------------------------------------- VideoLLaMA2.1 family

def test_oracle_v21_matches_reference_goldens(golden_small_v21):
    g = golden_small_v21
    cfg = g["cfg"]
    assert O.vision_family(cfg) == "siglip" and O.llm_family(cfg) == "qwen2" and O.conv3d_padding(cfg) == 0
    sd = O.seeded_state_dict(cfg, g["seed"], round_bf16=True)
    with torch.no_grad():
        assert torch.allclose(O.normalise_frames_u8_siglip(g["frames_u8"].numpy()), g["frames"], atol=1e-6)
        out, hs = O.siglip_tower(sd, cfg, g["frames"], return_hidden=True)
        for a, b in zip(hs, g["vit_hidden"]):
            assert torch.allclose(a, b, atol=2e-5, rtol=1e-5)
        assert torch.allclose(out, g["tower_out"], atol=2e-5, rtol=1e-5)
        feats, st = O.stc_connector(sd, out.view(1, *out.shape), return_stages=True, padding=0)
        assert torch.allclose(st["sampler"], g["stc_sampler"], atol=2e-5, rtol=1e-5)
        assert torch.allclose(feats, g["mm_features"], atol=2e-5, rtol=1e-5)
        assert feats.shape[1] == O.n_visual_tokens(4, 4, padding=0) == 8
        emb = O.splice_inputs_embeds(sd, g["input_ids"], [feats[0]])
        assert torch.allclose(emb, g["inputs_embeds"], atol=2e-5, rtol=1e-5)
        logits, _ = O.mistral_forward(sd, cfg, emb, last_only=False)
        assert torch.allclose(logits, g["prefill_logits"], atol=1e-4, rtol=1e-4)
        toks, step_logits = O.greedy_generate(sd, cfg, emb, 8)
        assert toks == g["new_tokens"].tolist()
        assert torch.allclose(step_logits, g["step_logits"], atol=1e-4, rtol=1e-4)
    assert O.n_visual_tokens(16, 27, padding=0) == 1352          # SURVEY 8f row 1 [probe]


@pytest.mark.skipif(not RH.reference_available(), reason="reference tree only exists in the build container")
def test_oracle_v21_matches_live_reference():
    cfg = O.config_small_v21(4)
    model, _ = RH.build_reference_model(cfg)
    RH.reseed_weights(model, 78)
    sd = O.seeded_state_dict(cfg, 78, round_bf16=False)
    ref_sd = {k: v for k, v in model.state_dict().items() if torch.is_floating_point(v)}
    assert set(ref_sd) == set(dict(O.state_dict_names(cfg)))
    rng = np.random.default_rng(6)
    fr = O.normalise_frames_u8_siglip(rng.integers(0, 256, (4, 56, 56, 3), dtype=np.uint8))
    with torch.no_grad():
        a = model.encode_images_or_videos([(fr, "video")])
        b = O.encode_images_or_videos(sd, cfg, [(fr, "video")])
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
        ids = torch.tensor([1] + list(rng.integers(3, 512, 5)) + [-201] + list(rng.integers(3, 512, 4)))
        out = model.generate(ids[None], attention_mask=torch.ones_like(ids)[None], images=[(fr, "video")], do_sample=False,
                             max_new_tokens=5, use_cache=True, pad_token_id=0, eos_token_id=None)
        toks, _ = O.generate(sd, cfg, ids, fr, 5)
        assert out[0].tolist() == toks
