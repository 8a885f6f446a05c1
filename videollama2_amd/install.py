"""Drop-in seams into the reference package (only meaningful where `videollama2` is importable).

    accelerate(model)   -- take a loaded reference `Videollama2MistralForCausalLM` or `Videollama2Qwen2ForCausalLM` (what
                           `model_init` returns, videollama2/__init__.py:14-29) and re-route its three seams to the HIP path:
                             model.get_model().vision_tower  -> HipCLIPVisionTower / HipSiglipVisionTower (encoder.py:154 seam)
                             model.get_model().mm_projector  -> HipSTCConnector (stc_connector / _v35)     (projector.py:95 seam)
                             model.generate                  -> HipMistralDecoder / HipQwen2Decoder loop
                                                                (videollama2_mistral.py:110, videollama2_qwen2.py:108 seams)
                           `mm_infer(tensor, instruct, model, tokenizer)` then runs unchanged.
    install()           -- wrap `videollama2.model_init` so every model it returns is accelerated.
The reference's modules are NOT kept as a fallback: after accelerate() the HF decoder layers are dropped."""
import types

import torch

from .config import check_supported, from_hf_config
from .model import VideoLLaMA2Hip


def accelerate(ref_model, device="cuda", max_seq_len=4096, free_reference_weights=True):
    hf_cfg = ref_model.config
    if getattr(hf_cfg, "mm_projector_type", None) not in ("stc_connector", "stc_connector_v35"):
        raise Exception(f"Unsupported projector type {getattr(hf_cfg, 'mm_projector_type', None)}!!!")
    if getattr(hf_cfg, "model_type", "") not in ("videollama2_mistral", "videollama2_qwen2"):
        raise ValueError(f"HIP path: model type {getattr(hf_cfg, 'model_type', None)} not built (videollama2_mistral, videollama2_qwen2)")
    tower = ref_model.get_vision_tower()
    if not any(t in type(tower).__name__.lower() for t in ("clip", "siglip")):
        raise ValueError(f"Unknown vision tower: {type(tower).__name__}")          # encoder.py:162
    cfg = from_hf_config(hf_cfg, tower.config)
    check_supported(cfg)
    sd = {k: v for k, v in ref_model.state_dict().items() if torch.is_floating_point(v)}
    hip = VideoLLaMA2Hip(cfg, sd, device, max_seq_len, image_processor=tower.image_processor)
    inner = ref_model.get_model()
    inner.vision_tower = hip.vision_tower
    inner.mm_projector = hip.mm_projector
    ref_model._vl2hip = hip

    def generate(self, inputs=None, images=None, **kwargs):
        if "eos_token_id" not in kwargs and getattr(self, "generation_config", None) is not None:
            kwargs["eos_token_id"] = self.generation_config.eos_token_id
        return self._vl2hip.generate(inputs, images=images, **kwargs)

    ref_model.generate = types.MethodType(generate, ref_model)
    if free_reference_weights:
        inner.layers = torch.nn.ModuleList()       # the HF decoder stack is replaced, not shadowed
    return ref_model


def install(device="cuda", max_seq_len=4096):
    import videollama2
    if getattr(videollama2.model_init, "_vl2hip_wrapped", False):
        return
    orig = videollama2.model_init

    def model_init(model_path=None, **kwargs):
        model, processor, tokenizer = orig(model_path, **kwargs)
        return accelerate(model, device, max_seq_len), processor, tokenizer

    model_init._vl2hip_wrapped = True
    videollama2.model_init = model_init
