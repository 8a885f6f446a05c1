"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (videollama2_amd/).

Imports the *real* reference package in place from /root/reference (read-only, never copied) with the
four shims SURVEY.md section 8(c) lists, and builds `Videollama2MistralForCausalLM` from a config dict with
seeded random weights.  It exists only in this container: /root/reference is absent on the GPU box, so
nothing under tests/ that is marked `gpu`, `bench.py` or `smoke()` may call into this file.  Its two users
are `oracle/make_golden.py` (mints tests/golden/*.pt) and the CPU test that pins `oracle/vl2_oracle.py`
(the restatement that does travel) against the reference itself.

Shims (reference file:line that needs them):
  1. timm                -> oracle/shims/timm        (videollama2/model/projector.py:22-23)
  2. decord/cv2/imageio  -> oracle/shims/*           (videollama2/mm_utils.py:8-13)
  3. transformers.TRANSFORMERS_CACHE                 (videollama2/model/projector.py:24)
  4. flash_attention_2 -> eager                      (videollama2/model/encoder.py:24)
"""
import json
import os
import sys
import tempfile

import torch

REFERENCE_ROOT = "/root/reference"
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "videollama2"))


_ref = None


def import_reference():
    """Import /root/reference/videollama2 in place.  Returns the package module."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError("reference tree not present (expected on the GPU box); goldens are in tests/golden/")
    # every other `from transformers import ...` FIRST (lazy-module gotcha, SURVEY 8c shim 3)
    import transformers  # noqa: F401  (imported for its side effect: the lazy module must be initialised first)
    from transformers import (CLIPVisionModel, CLIPImageProcessor, CLIPVisionConfig, SiglipVisionModel,  # noqa
                              SiglipImageProcessor, SiglipVisionConfig, MistralConfig, MistralModel,
                              MistralForCausalLM, AutoConfig, AutoModelForCausalLM, PretrainedConfig,
                              StoppingCriteria, AutoTokenizer, BitsAndBytesConfig)
    for p in (_SHIMS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.modules["transformers"].TRANSFORMERS_CACHE = os.path.join(tempfile.gettempdir(), "hf_cache_unused")
    import videollama2  # noqa: E402  (the reference, unmodified)
    import videollama2.model.encoder as enc

    real_clip = enc.CLIPVisionModel

    class _EagerCLIPVisionModel(real_clip):  # shim 4: encoder.py:24 hard-codes flash_attention_2
        def __init__(self, config, *a, **k):
            config._attn_implementation = "eager"
            super().__init__(config, *a, **k)

    enc.CLIPVisionModel = _EagerCLIPVisionModel
    real_siglip = enc.SiglipVisionModel

    class _EagerSiglipVisionModel(real_siglip):  # shim 4 again: encoder.py:97 hard-codes flash_attention_2 for SigLIP too
        def __init__(self, config, *a, **k):
            config._attn_implementation = "eager"
            super().__init__(config, *a, **k)

    enc.SiglipVisionModel = _EagerSiglipVisionModel
    _ref = videollama2
    return _ref


def write_clip_dir(cfg, root=None):
    """A directory whose path contains 'clip' (encoder.py:157 string match) holding config.json and
    preprocessor_config.json, because CLIPVisionTower.__init__ calls from_pretrained for both (encoder.py:21-23)."""
    v = cfg["vision"]
    root = root or tempfile.mkdtemp(prefix="vl2_clip_")
    d = os.path.join(root, "clip-vit-synthetic")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(model_type="clip_vision_model", hidden_size=v["hidden_size"],
                       intermediate_size=v["intermediate_size"], num_hidden_layers=v["num_hidden_layers"],
                       num_attention_heads=v["num_attention_heads"], image_size=v["image_size"],
                       patch_size=v["patch_size"], hidden_act="quick_gelu", layer_norm_eps=v["layer_norm_eps"],
                       projection_dim=v["hidden_size"], num_channels=3), f)
    with open(os.path.join(d, "preprocessor_config.json"), "w") as f:
        json.dump(dict(image_processor_type="CLIPImageProcessor", do_resize=True, size={"shortest_edge": v["image_size"]},
                       resample=3, do_center_crop=True, crop_size={"height": v["image_size"], "width": v["image_size"]},
                       do_rescale=True, rescale_factor=1 / 255, do_normalize=True, do_convert_rgb=True,
                       image_mean=[0.48145466, 0.4578275, 0.40821073],
                       image_std=[0.26862954, 0.26130258, 0.27577711]), f)
    return d


def write_siglip_dir(cfg, root=None):
    """Same for SiglipVisionTower (encoder.py:94-96): a path containing 'siglip' with config.json + preprocessor_config.json
    (public google/siglip-so400m-patch14-384 preprocessor values: plain resize to image_size^2, mean = std = 0.5)."""
    v = cfg["vision"]
    root = root or tempfile.mkdtemp(prefix="vl2_siglip_")
    d = os.path.join(root, "siglip-synthetic")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(model_type="siglip_vision_model", hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"],
                       num_hidden_layers=v["num_hidden_layers"], num_attention_heads=v["num_attention_heads"],
                       image_size=v["image_size"], patch_size=v["patch_size"], hidden_act="gelu_pytorch_tanh",
                       layer_norm_eps=v["layer_norm_eps"], num_channels=3), f)
    with open(os.path.join(d, "preprocessor_config.json"), "w") as f:
        json.dump(dict(image_processor_type="SiglipImageProcessor", do_resize=True, size={"height": v["image_size"], "width": v["image_size"]},
                       resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True, do_convert_rgb=None,
                       image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5]), f)
    return d


def build_reference_model_qwen2(cfg, seed=1234, tower_dir=None):
    """Videollama2Qwen2ForCausalLM(config) with a SigLIP tower and stc_connector_v35 (the VideoLLaMA2.1 family); fp32, eval."""
    ref = import_reference()
    from videollama2.model.videollama2_qwen2 import Videollama2Qwen2ForCausalLM, Videollama2Qwen2Config
    l = cfg["llm"]
    tower_dir = tower_dir or write_siglip_dir(cfg)
    hf_cfg = Videollama2Qwen2Config(
        hidden_size=l["hidden_size"], intermediate_size=l["intermediate_size"], num_hidden_layers=l["num_hidden_layers"],
        num_attention_heads=l["num_attention_heads"], num_key_value_heads=l["num_key_value_heads"], vocab_size=l["vocab_size"],
        rms_norm_eps=l["rms_norm_eps"], rope_theta=l["rope_theta"], max_position_embeddings=32768, use_sliding_window=False,
        tie_word_embeddings=False, attn_implementation="eager",
        mm_vision_tower=tower_dir, mm_projector_type=cfg.get("projector", "stc_connector_v35"),
        mm_hidden_size=cfg["vision"]["hidden_size"], mm_vision_select_layer=cfg["vision"]["select_layer"],
        mm_vision_select_feature="patch", num_frames=cfg["num_frames"], bos_token_id=1, eos_token_id=2, pad_token_id=0)
    torch.manual_seed(seed)
    model = Videollama2Qwen2ForCausalLM(hf_cfg).float().eval()
    return model, ref


def build_reference_model(cfg, seed=1234, clip_dir=None):
    """Videollama2MistralForCausalLM(config) as SURVEY 8(c) 'Local files needed' describes; fp32, eval, CPU."""
    if cfg["llm"].get("family", "mistral") == "qwen2":
        return build_reference_model_qwen2(cfg, seed, clip_dir)
    ref = import_reference()
    from videollama2.model.videollama2_mistral import Videollama2MistralForCausalLM, Videollama2MistralConfig
    l = cfg["llm"]
    clip_dir = clip_dir or write_clip_dir(cfg)
    hf_cfg = Videollama2MistralConfig(
        hidden_size=l["hidden_size"], intermediate_size=l["intermediate_size"],
        num_hidden_layers=l["num_hidden_layers"], num_attention_heads=l["num_attention_heads"],
        num_key_value_heads=l["num_key_value_heads"], vocab_size=l["vocab_size"], rms_norm_eps=l["rms_norm_eps"],
        rope_theta=l["rope_theta"], head_dim=l["head_dim"], max_position_embeddings=32768, sliding_window=None,
        tie_word_embeddings=False, attn_implementation="eager",
        mm_vision_tower=clip_dir, mm_projector_type="stc_connector", mm_hidden_size=cfg["vision"]["hidden_size"],
        mm_vision_select_layer=cfg["vision"]["select_layer"], mm_vision_select_feature="patch",
        num_frames=cfg["num_frames"], bos_token_id=1, eos_token_id=2, pad_token_id=0)
    torch.manual_seed(seed)
    model = Videollama2MistralForCausalLM(hf_cfg).float().eval()
    return model, ref


def reseed_weights(model, seed=1234, lm_head_scale=1.0):
    """Deterministic, layout-independent re-initialisation used for goldens: every parameter is drawn from its
    own generator keyed by its NAME, so the same tensors can be rebuilt anywhere (GPU box included) without
    the reference.  Matrices ~ N(0, 1/fan_in)-ish so activations stay O(1) through the depth; norm weights
    ~ 1 + 0.1 N(0,1); biases ~ 0.02 N(0,1).  See oracle/vl2_oracle.py:seeded_state_dict (same rule)."""
    from .vl2_oracle import seeded_tensor
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            new[k] = v
            continue
        new[k] = seeded_tensor(k, tuple(v.shape), seed, lm_head_scale)
    model.load_state_dict(new, strict=True)
    return model
