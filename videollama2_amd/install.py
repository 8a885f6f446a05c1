"""Drop-in seams into the reference package (only meaningful where `videollama2` is importable).

    accelerate(model)   -- take a loaded reference `Videollama2MistralForCausalLM` or `Videollama2Qwen2ForCausalLM` (what
                           `model_init` returns, videollama2/__init__.py:14-29) and re-route its three seams to the HIP path:
                             model.get_model().vision_tower  -> HipCLIPVisionTower / HipSiglipVisionTower (encoder.py:154 seam)
                             model.get_model().mm_projector  -> HipSTCConnector (stc_connector / _v35)     (projector.py:95 seam)
                             model.generate                  -> HipMistralDecoder / HipQwen2Decoder loop
                                                                (videollama2_mistral.py:110, videollama2_qwen2.py:108 seams)
                           `mm_infer(tensor, instruct, model, tokenizer)` then runs unchanged.
    install()           -- patch the reference's FACTORIES and loader classes (build_vision_tower, build_vision_projector, the
                           classes `load_pretrained_model` loads, the VLLMs registry): models are BUILT on the HIP path, no HF
                           tower / projector / decoder is constructed first.
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


class HipCausalLMLoader:
    """Stands in for `Videollama2MistralForCausalLM` / `Videollama2Qwen2ForCausalLM` at the ONE place the reference's inference
    entry uses them: `<Class>.from_pretrained(model_path, low_cpu_mem_usage=True, config=config, **kwargs)` inside
    `load_pretrained_model` (videollama2/model/__init__.py:157-165, reached from `model_init`, videollama2/__init__.py:14-29).
    Returns the HIP model built straight from the checkpoint files -- no HF tower / projector / decoder is ever constructed or
    loaded onto the GPU first.  The object it returns carries what `model_init` / `mm_infer` touch: `get_vision_tower()`
    (`.image_processor`), `.config` (`model_type`, `num_frames`), `.generate(input_ids, images=..., **hf_kwargs)`, `.eval()`,
    `.to()`, `.device`."""
    device, max_seq_len = "cuda", 4096
    _orig = None          # the reference class this loader replaced (install() sets it): plain-LLM checkpoints are handed back to it

    @classmethod
    def from_pretrained(cls, model_path, *args, config=None, **kwargs):
        from . import api
        from .tower import default_image_processor, default_siglip_image_processor
        # The LoRA / `model_base` / pretrain-projector branches of load_pretrained_model (model/__init__.py:70-156) call
        # `<Class>.from_pretrained(model_base, ...)` on a PLAIN language-model checkpoint, then touch `model.lm_head`, load
        # `mm_projector` weights into it and wrap it with PeftModel: that is the reference class's job.  A checkpoint without the
        # multimodal config keys goes back to the class this loader replaced; `model_init` accelerates the finished model afterwards.
        import json
        import os
        cj = os.path.join(str(model_path), "config.json")
        hfj = None
        if os.path.isfile(cj):
            with open(cj) as fh:
                hfj = json.load(fh)
        elif not os.path.isdir(str(model_path)):                      # a hub id: resolve its config.json through the hub cache
            try:
                from huggingface_hub import hf_hub_download
                with open(hf_hub_download(str(model_path), "config.json")) as fh:
                    hfj = json.load(fh)
            except Exception as e:                                    # offline / unknown id: the reference class decides, and says so
                import warnings
                warnings.warn(f"HIP path: config.json of '{model_path}' is not readable ({type(e).__name__}); handing the checkpoint to "
                              f"{getattr(cls._orig, '__name__', 'the reference class')} -- it will NOT run on the HIP kernels")
        # delegate only on evidence: a readable config WITHOUT the multimodal keys (or no readable config at all, warned about above)
        if cls._orig is not None and (hfj is None or not (hfj.get("mm_vision_tower") and hfj.get("mm_projector_type"))):
            return cls._orig.from_pretrained(model_path, *args, config=config, **kwargs)
        for k in ("load_in_4bit", "load_in_8bit", "quantization_config"):          # model/__init__.py:57-69: bitsandbytes loading
            if kwargs.get(k):
                raise NotImplementedError(f"HIP path: `{k}` (bitsandbytes quantised loading) is not built; load the bf16 checkpoint")
        cfg, hf = api.config_from_checkpoint(model_path)
        check_supported(cfg)
        siglip = cfg["vision"]["family"] == "siglip"
        proc = (default_siglip_image_processor if siglip else default_image_processor)(cfg["vision"]["image_size"])
        dm = kwargs.get("device_map")
        dev = dm[""] if isinstance(dm, dict) and "" in dm else cls.device
        model = VideoLLaMA2Hip(cfg, api.load_state_dict(model_path), dev, cls.max_seq_len, image_processor=proc)
        model.config = config if config is not None else types.SimpleNamespace(**hf)
        return model


def install(device="cuda", max_seq_len=4096, patch_factories=True, patch_loader=True):
    """Re-route the reference package (must be importable) to the HIP path at its own seams:
      * `build_vision_tower` (encoder.py:154-164)  -> lazy.LazyHipVisionTower   (CLIP and SigLIP towers)
      * `build_vision_projector` (projector.py:95-122) -> lazy.LazyHipSTCConnector for stc_connector / stc_connector_v35
        (other projector types fall through to the reference's own factory)
        -- both in the modules that DEFINE them and in videollama2_arch, which imported the names (arch.py:25-26), so every
        `Videollama2*ForCausalLM(config)` built afterwards hosts HIP modules with the reference's state-dict keys;
      * the loader classes `load_pretrained_model` calls `.from_pretrained` on, and the `VLLMs` registry
        (model/__init__.py:31-37, :157-165) -> HipCausalLMLoader for model types videollama2 / videollama2_mistral /
        videollama2_qwen2: `videollama2.model_init(path)` then returns the HIP model without building any HF module;
      * `videollama2.model_init` itself keeps working for model objects that were built some other way (wrapped with
        `accelerate`, as before).
    Idempotent."""
    import videollama2
    import videollama2.model as vm
    import videollama2.model.encoder as enc
    import videollama2.model.projector as proj
    import videollama2.model.videollama2_arch as arch
    from .lazy import LazyHipSTCConnector, LazyHipVisionTower
    if getattr(videollama2, "_vl2hip_installed", False):
        return
    saved = dict(enc_tower=enc.build_vision_tower, arch_tower=arch.build_vision_tower, proj_proj=proj.build_vision_projector,
                 arch_proj=arch.build_vision_projector, mistral=vm.Videollama2MistralForCausalLM, qwen2=vm.Videollama2Qwen2ForCausalLM,
                 vllms=dict(vm.VLLMs), model_init=videollama2.model_init)
    videollama2._vl2hip_saved = saved
    if patch_factories:
        ref_build_projector = proj.build_vision_projector

        def build_vision_tower(vision_tower_cfg, **kwargs):                        # encoder.py:154-164, same dispatch and error
            name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
            if name is None or not ("clip" in name or "siglip" in name):
                raise ValueError(f"Unknown vision tower: {name}")
            kwargs.pop("load_pretrained", None)
            return LazyHipVisionTower(name, args=vision_tower_cfg, device=None, **kwargs)

        def build_vision_projector(config, delay_load=False, **kwargs):           # projector.py:95-122
            if getattr(config, "mm_projector_type", "linear") in ("stc_connector", "stc_connector_v35"):
                return LazyHipSTCConnector(config)
            return ref_build_projector(config, delay_load=delay_load, **kwargs)

        for mod in (enc, arch):
            mod.build_vision_tower = build_vision_tower
        for mod in (proj, arch):
            mod.build_vision_projector = build_vision_projector
    if patch_loader:
        loaders = {}
        for name in ("Videollama2MistralForCausalLM", "Videollama2Qwen2ForCausalLM"):
            loaders[name] = type("HipCausalLMLoader", (HipCausalLMLoader,), dict(device=device, max_seq_len=max_seq_len, _orig=getattr(vm, name)))
            setattr(vm, name, loaders[name])
        for key, name in (("videollama2", "Videollama2MistralForCausalLM"), ("videollama2_mistral", "Videollama2MistralForCausalLM"),
                          ("videollama2_qwen2", "Videollama2Qwen2ForCausalLM")):
            vm.VLLMs[key] = loaders[name]
    orig = videollama2.model_init

    def model_init(model_path=None, **kwargs):
        model, processor, tokenizer = orig(model_path, **kwargs)
        if not isinstance(model, VideoLLaMA2Hip) and not hasattr(model, "_vl2hip"):
            model = accelerate(model, device, max_seq_len)
        return model, processor, tokenizer

    videollama2.model_init = model_init
    videollama2._vl2hip_installed = True


def uninstall():
    """Undo `install()` (tests; A/B against the unmodified reference in one process)."""
    import videollama2
    import videollama2.model as vm
    import videollama2.model.encoder as enc
    import videollama2.model.projector as proj
    import videollama2.model.videollama2_arch as arch
    sv = getattr(videollama2, "_vl2hip_saved", None)
    if not getattr(videollama2, "_vl2hip_installed", False) or sv is None:
        return
    enc.build_vision_tower, arch.build_vision_tower = sv["enc_tower"], sv["arch_tower"]
    proj.build_vision_projector, arch.build_vision_projector = sv["proj_proj"], sv["arch_proj"]
    vm.Videollama2MistralForCausalLM, vm.Videollama2Qwen2ForCausalLM = sv["mistral"], sv["qwen2"]
    vm.VLLMs.clear()
    vm.VLLMs.update(sv["vllms"])
    videollama2.model_init = sv["model_init"]
    videollama2._vl2hip_installed = False
