"""The inference API surface and the plug-in seams ON THE GPU:
  * `api.model_init(local checkpoint dir)` + `api.mm_infer(...)` (videollama2/__init__.py:14-114) on a synthetic safetensors
    checkpoint of both families, tokens checked against the golden ones minted from the real reference;
  * the factory-level drop-ins `install.install()` plants into the reference (lazy.LazyHipVisionTower / LazyHipSTCConnector):
    state-dict keys of the reference modules, strict load, first-call packing, output identical to the directly built HIP modules;
  * the padded-batch path of generate (videollama2_arch.py:227-261) and `forward(..., images=)` (videollama2_mistral.py:63-108)."""
import types

import pytest
import torch

from oracle import vl2_oracle as O
from tests.util import ToyTokenizer, rel, write_synthetic_checkpoint

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("family", ["v2", "v21"])
def test_model_init_and_mm_infer_on_device(golden_small, golden_small_v21, family, tmp_path):
    from videollama2_amd import api
    g = golden_small if family == "v2" else golden_small_v21
    cfg = g["cfg"]
    write_synthetic_checkpoint(tmp_path, cfg, g["seed"], O)
    tok = ToyTokenizer(cfg["llm"]["vocab_size"])
    model, processor, tok2 = api.model_init(str(tmp_path), device=DEV, max_seq_len=192, tokenizer=tok)
    assert tok2.pad_token == "<unk>" and set(processor) == {"image", "video"} and model.device.type == torch.device(DEV).type
    frames = processor["video"](g["frames_u8"].numpy())
    assert torch.allclose(frames, g["frames"], atol=1e-6)
    # the golden request through the model object model_init returned: tokens of the real reference
    ids = g["input_ids"][None].to(DEV)
    out = model.generate(ids, attention_mask=torch.ones_like(ids), images=[(g["frames"].to(DEV), "video")], do_sample=False, max_new_tokens=4)
    assert out[0].tolist() == g["new_tokens"][:4].tolist()
    # mm_infer end to end (prompt template -> sentinel ids -> HIP generate (hipGraph decode) -> text), eager decode as the cross-check
    text = api.mm_infer(frames, "what happens in the clip ?", model, tok, modal="video", max_new_tokens=6)
    prompt = tok.apply_chat_template(tok.prompts[-1])
    pids = api.tokenizer_multimodal_token(prompt, tok, "<video>", return_tensors="pt")[None].to(DEV)
    ref = model.generate(pids, attention_mask=torch.ones_like(pids), images=[(frames.half().to(DEV), "video")], do_sample=False,
                         max_new_tokens=6, eos_token_id=2, use_graph=False)
    assert len(text) > 0 and text == tok.batch_decode(ref)[0].strip()
    roles = [m["role"] for m in tok.prompts[-1]]
    assert roles == (["system", "user"] if family == "v2" else ["user"])            # __init__.py:72-83
    from PIL import Image
    Image.fromarray(g["frames_u8"][0].numpy()).save(str(tmp_path / "frame0.png"))      # process_image takes a path (mm_utils.py:25-40)
    img = api.mm_infer(processor["image"](str(tmp_path / "frame0.png")), "describe", model, tok, modal="image", max_new_tokens=3)
    assert isinstance(img, str) and len(img) > 0
    with pytest.raises(ValueError, match="Unsupported modal"):
        api.mm_infer(frames, "x", model, tok, modal="audio")


@pytest.mark.parametrize("family", ["v2", "v21"])
def test_factory_drop_ins_load_reference_state_dict_and_match_direct_modules(golden_small, golden_small_v21, family):
    """What the patched build_vision_tower / build_vision_projector return (no reference needed to build them): keys of the
    reference modules, strict load_state_dict, pack on first call, same output as the directly constructed HIP modules."""
    from videollama2_amd.connector import HipSTCConnector
    from videollama2_amd.lazy import LazyHipSTCConnector, LazyHipVisionTower
    from videollama2_amd.tower import HipCLIPVisionTower, HipSiglipVisionTower
    g = golden_small if family == "v2" else golden_small_v21
    cfg = g["cfg"]
    v = cfg["vision"]
    sd = O.seeded_state_dict(cfg, g["seed"])
    import json
    import os
    import tempfile
    d = os.path.join(tempfile.mkdtemp(), "siglip-synthetic" if O.vision_family(cfg) == "siglip" else "clip-synthetic")
    os.makedirs(d)
    json.dump(dict(v, model_type="siglip_vision_model" if O.vision_family(cfg) == "siglip" else "clip_vision_model"), open(os.path.join(d, "config.json"), "w"))
    args = types.SimpleNamespace(mm_vision_tower=d, mm_vision_select_layer=v["select_layer"], mm_vision_select_feature="patch",
                                 mm_projector_type=cfg.get("projector", "stc_connector"), mm_hidden_size=v["hidden_size"],
                                 hidden_size=cfg["llm"]["hidden_size"])
    host = torch.nn.Module()
    host.vision_tower = LazyHipVisionTower(d, args)
    host.mm_projector = LazyHipSTCConnector(args)
    want = {k[len("model."):] for k in sd if k.startswith(("model.vision_tower.", "model.mm_projector."))}
    have = set(host.state_dict().keys())
    assert want == have, (sorted(want - have)[:5], sorted(have - want)[:5])       # exactly the reference's keys (pinned vs the real model in test_oracle_pin)
    host.load_state_dict({k[len("model."):]: t.bfloat16() for k, t in sd.items() if k.startswith(("model.vision_tower.", "model.mm_projector."))}, strict=True)
    host.to(DEV)
    frames = g["frames"].to(DEV)
    tower_out = host.vision_tower(frames)
    feats = host.mm_projector(tower_out.view(1, *tower_out.shape))
    direct_t = (HipSiglipVisionTower if O.vision_family(cfg) == "siglip" else HipCLIPVisionTower)(cfg, sd, DEV)
    direct_c = HipSTCConnector(sd, DEV, padding=O.conv3d_padding(cfg))
    assert torch.equal(tower_out, direct_t(frames))
    assert torch.equal(feats, direct_c(tower_out.view(1, *tower_out.shape)))
    assert rel(feats, g["mm_features"]) < 2.5e-2
    assert host.vision_tower.vision_tower is None and not list(host.mm_projector.parameters())     # hosts released after packing


def test_padded_batch_generate_and_forward_with_images_on_device(golden_small):
    """arch.py:227-261 + videollama2_mistral.py:63-108: right-padded batch of two video prompts through generate (each row = the
    tokens the sequence gets alone, batched decode under a hipGraph), and forward(input_ids, images=) logits vs the golden ones."""
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    sd = O.seeded_state_dict(cfg, g["seed"])
    m = VideoLLaMA2Hip(cfg, sd, DEV, max_seq_len=96)
    idsA = g["input_ids"]
    idsB = torch.cat([idsA[:3], idsA[6:]])
    L = idsA.numel()
    batch = torch.zeros((2, L), dtype=torch.long)
    batch[0] = idsA
    batch[1, :idsB.numel()] = idsB
    mask = torch.ones_like(batch)
    mask[1, idsB.numel():] = 0
    fr2 = torch.flip(g["frames"], dims=[0]).contiguous()
    images = [(g["frames"].to(DEV), "video"), (fr2.to(DEV), "video")]
    alone = [m.generate(idsA[None].to(DEV), attention_mask=torch.ones(1, L, dtype=torch.long), images=images[:1], max_new_tokens=4)[0].tolist(),
             m.generate(idsB[None].to(DEV), attention_mask=torch.ones(1, idsB.numel(), dtype=torch.long), images=images[1:], max_new_tokens=4)[0].tolist()]
    assert alone[0] == g["new_tokens"][:4].tolist()
    out = m.generate(batch.to(DEV), attention_mask=mask.to(DEV), images=images, max_new_tokens=4, pad_token_id=0)
    assert out.tolist() == alone
    res = m(input_ids=idsA[None].to(DEV), attention_mask=torch.ones(1, L, dtype=torch.long), images=images[:1])
    assert rel(res.logits[0], g["prefill_logits"]) < 2.5e-2


def test_continuous_batching_on_device_equals_solo_decode(golden_small):
    """serving.ContinuousBatcher on the GPU, one captured hipGraph per occupancy: staggered admission, early retirement and slot
    reuse; every request's tokens equal its solo greedy decode (hipGraph) -- SURVEY 8f row 4, serve/model_worker.py:263-300."""
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"]), DEV, max_seq_len=128)
    idsA = g["input_ids"].to(DEV)
    variants = [idsA, torch.cat([idsA[:3], idsA[6:]]), torch.cat([idsA[:2], idsA[5:]]), idsA[:4], torch.cat([idsA[:4], idsA[8:]])]
    fr = [g["frames"].to(DEV), torch.flip(g["frames"], dims=[0]).contiguous().to(DEV)]
    reqs = [(variants[0], [(fr[0], "video")], 9), (variants[1], [(fr[1], "video")], 3), (variants[2], [(fr[0], "video")], 6),
            (variants[3], None, 4), (variants[4], [(fr[1], "video")], 7)]
    solo = [m.generate(ids[None], attention_mask=torch.ones(1, ids.numel(), dtype=torch.long, device=DEV), images=im, max_new_tokens=n)[0].tolist()
            for ids, im, n in reqs]
    assert solo[0][:8] == g["new_tokens"][:8].tolist()
    for use_graph in (True, False):
        b = m.batcher(max_slots=3, use_graph=use_graph)
        rid = [b.submit(reqs[i][0], reqs[i][1], max_new_tokens=reqs[i][2]) for i in range(2)]
        b.step()
        rid += [b.submit(reqs[i][0], reqs[i][1], max_new_tokens=reqs[i][2]) for i in range(2, 4)]     # one joins now, one waits for a slot
        b.step(); b.step()
        rid.append(b.submit(reqs[4][0], reqs[4][1], max_new_tokens=reqs[4][2]))
        done = b.run()
        assert [done[r].tolist() for r in rid] == solo, use_graph
        assert b.inner.steps < sum(n for _, _, n in reqs)          # the steps were shared
