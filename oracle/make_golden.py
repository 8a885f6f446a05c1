"""TEST INFRASTRUCTURE ONLY.  Mints tests/golden/*.pt by running the REAL reference (imported in place from
/root/reference via oracle/ref_harness.py) on seeded synthetic inputs.  Run in the build container only:

    python -m oracle.make_golden

The fixtures travel to the GPU box (the reference does not).  Weights are NOT stored: they are rebuilt from
oracle.vl2_oracle.seeded_state_dict(cfg, seed) (name-keyed generators), rounded once to bf16.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as RH  # noqa: E402
from oracle import vl2_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 1234


def mint_small(T=4, n_new=8, cfg=None, fname="small_T4.pt"):
    cfg = cfg or O.config_small(T)
    model, ref = RH.build_reference_model(cfg)
    RH.reseed_weights(model, SEED)
    # round weights once to bf16 (what the HIP path stores), keep fp32 math
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    from videollama2.mm_utils import process_video, frame_sample
    proc = model.get_vision_tower().image_processor
    rng = np.random.default_rng(0)
    S = cfg["vision"]["image_size"]
    frames_u8 = rng.integers(0, 256, (T, S, S, 3), dtype=np.uint8)
    frames = process_video(frames_u8, proc, aspect_ratio=None, num_frames=T)          # mm_utils.py:179-201
    g = dict(cfg=cfg, seed=SEED, frames_u8=torch.from_numpy(frames_u8), frames=frames)

    # non-square, fewer frames than num_frames (black-frame padding mm_utils.py:190-191), bicubic resize + crop
    odd_u8 = rng.integers(0, 256, (3, 70, 90, 3), dtype=np.uint8)
    g["odd_u8"] = torch.from_numpy(odd_u8)
    g["odd_frames"] = process_video([f for f in odd_u8], proc, aspect_ratio=None, num_frames=T)
    g["odd_frames_pad"] = process_video(odd_u8, proc, aspect_ratio="pad", num_frames=T)
    g["frame_sample"] = {(d, n): torch.from_numpy(frame_sample(d, mode="uniform", num_frames=n))
                         for d, n in ((8, 8), (17, 8), (100, 16), (301, 32), (33, 16), (2, 8), (1000, 8))}

    tower = model.get_vision_tower()
    stages = {}
    mp = model.get_model().mm_projector
    hooks = [mp.s1.register_forward_hook(lambda m, i, o: stages.__setitem__("s1", o.detach().clone())),
             mp.sampler.register_forward_hook(lambda m, i, o: stages.__setitem__("sampler", o.detach().clone())),
             mp.s2.register_forward_hook(lambda m, i, o: stages.__setitem__("s2", o.detach().clone()))]
    with torch.no_grad():
        hs = tower.vision_tower(frames, output_hidden_states=True).hidden_states
        g["vit_hidden"] = [h.clone() for h in hs[:cfg["vision"]["num_hidden_layers"]]]   # 0 .. L-1 (L-1 is selected)
        g["tower_out"] = tower(frames)                                                      # encoder.py:41-53
        g["mm_features"] = model.encode_images_or_videos([(frames, "video")])               # arch.py:114-134
        g["stc_s1"], g["stc_sampler"], g["stc_s2"] = stages["s1"], stages["sampler"], stages["s2"]
        for h in hooks:
            h.remove()
        V = cfg["llm"]["vocab_size"]
        ids = torch.tensor([1] + list(rng.integers(3, V, 7)) + [-201] + list(rng.integers(3, V, 9)))
        g["input_ids"] = ids
        _, mask, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids[None], torch.ones_like(ids)[None],
                                                                        None, None, [(frames, "video")])
        g["inputs_embeds"] = emb[0]
        g["attention_mask"] = mask[0]
        g["prefill_logits"] = model(inputs_embeds=emb, attention_mask=mask).logits[0]
        out = model.generate(ids[None], attention_mask=torch.ones_like(ids)[None], images=[(frames, "video")],
                             do_sample=False, max_new_tokens=n_new, use_cache=True, pad_token_id=0, eos_token_id=None,
                             output_scores=True, return_dict_in_generate=True)
        g["new_tokens"] = out.sequences[0]
        g["step_logits"] = torch.stack([s[0] for s in out.scores])
    os.makedirs(OUT, exist_ok=True)
    torch.save(g, os.path.join(OUT, fname))
    print("wrote", fname, ";", {k: (tuple(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in g.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["v2", "v21"]
    if "v2" in which:
        mint_small()
    if "v21" in which:      # VideoLLaMA2.1 family: SigLIP tower + stc_connector_v35 + Qwen2
        mint_small(cfg=O.config_small_v21(4), fname="small_v21_T4.pt")
