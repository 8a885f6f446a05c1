"""Standalone inference API with the reference's shape (videollama2/__init__.py:14-114), for boxes WITHOUT the reference
package: `model_init(local_checkpoint_dir)` -> (model, processor, tokenizer) and `mm_infer(tensor, instruct, model, tokenizer,
modal)` -> str.  Where the reference package is installed, `install.accelerate(model)` keeps its own `model_init` / `mm_infer`.

`model_init` reads a LOCAL checkpoint directory (config.json + *.safetensors or pytorch_model*.bin, HF layout; no hub access on
the target box).  The vision tower's hyper-parameters come from `<dir>/vision_config.json` or a local tower directory when
present, otherwise from the public values of the two towers the released checkpoints use
(openai/clip-vit-large-patch14-336, google/siglip-so400m-patch14-384)."""
import copy
import glob
import json
import os
import types
from functools import partial

import torch

from .config import check_supported
from .constants import DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN, NUM_FRAMES
from .mm_utils import KeywordsStoppingCriteria, process_image, process_video, tokenizer_multimodal_token
from .model import VideoLLaMA2Hip
from .tower import default_image_processor, default_siglip_image_processor

_PUBLIC_TOWERS = {
    "clip-vit-large-patch14-336": dict(family="clip", hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                       num_attention_heads=16, image_size=336, patch_size=14, layer_norm_eps=1e-5),
    "siglip-so400m-patch14-384": dict(family="siglip", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                      num_attention_heads=16, image_size=384, patch_size=14, layer_norm_eps=1e-6),
}

# videollama2/__init__.py:72-81: the llama-2 style system prompt the reference prepends for these model types only
_SYSTEM_TYPES = ("videollama2", "videollama2_mistral", "videollama2_mixtral")
_SYSTEM_PROMPT = (
    "<<SYS>>\nYou are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.  "
    "Your answers should not include any harmful, unethical, racist, sexist, toxic, dangerous, or illegal content. Please ensure "
    "that your responses are socially unbiased and positive in nature.\n"
    "If a question does not make any sense, or is not factually coherent, explain why instead of answering something not "
    "correct. If you don't know the answer to a question, please don't share false information.\n<</SYS>>")


def _vision_config(hf, model_path):
    local = os.path.join(model_path, "vision_config.json")
    tower = str(hf.get("mm_vision_tower", ""))
    if os.path.isfile(local):
        v = json.load(open(local))
    elif os.path.isfile(os.path.join(tower, "config.json")):
        v = json.load(open(os.path.join(tower, "config.json")))
    else:
        key = tower.strip("/").split("/")[-1]
        if key not in _PUBLIC_TOWERS:
            raise ValueError(f"Unknown vision tower: {tower}")                       # encoder.py:162
        v = dict(_PUBLIC_TOWERS[key])
    fam = v.get("family") or ("siglip" if "siglip" in (str(v.get("model_type", "")) + tower).lower() else "clip")
    return dict(family=fam, hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"],
                num_hidden_layers=v["num_hidden_layers"], num_attention_heads=v["num_attention_heads"],
                image_size=v["image_size"], patch_size=v["patch_size"], layer_norm_eps=v.get("layer_norm_eps", 1e-5),
                select_layer=hf.get("mm_vision_select_layer", -2))


def config_from_checkpoint(model_path):
    """config.json of a VideoLLaMA2 checkpoint (keys of videollama2_arch.py:49-68 + the HF Mistral / Qwen2 config) -> cfg dict."""
    hf = json.load(open(os.path.join(model_path, "config.json")))
    mtype = hf.get("model_type", "videollama2_mistral")
    if mtype not in ("videollama2_mistral", "videollama2_qwen2"):
        raise ValueError(f"HIP path: model type {mtype} not built (videollama2_mistral, videollama2_qwen2)")
    rope = hf.get("rope_theta") or (hf.get("rope_parameters") or {}).get("rope_theta") or 1e6
    l = dict(family="qwen2" if "qwen2" in mtype else "mistral", hidden_size=hf["hidden_size"], intermediate_size=hf["intermediate_size"],
             num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=hf["num_attention_heads"],
             num_key_value_heads=hf.get("num_key_value_heads", hf["num_attention_heads"]),
             head_dim=hf.get("head_dim") or hf["hidden_size"] // hf["num_attention_heads"], vocab_size=hf["vocab_size"],
             rms_norm_eps=hf.get("rms_norm_eps", 1e-5), rope_theta=float(rope))
    cfg = dict(vision=_vision_config(hf, model_path), llm=l, projector=hf.get("mm_projector_type", "stc_connector"),
               num_frames=hf.get("num_frames", NUM_FRAMES))
    return cfg, hf


def load_state_dict(model_path):
    sd = {}
    st = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f))
    else:
        for f in sorted(glob.glob(os.path.join(model_path, "pytorch_model*.bin"))):
            sd.update(torch.load(f, map_location="cpu", weights_only=True))
    if not sd:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_path}")
    return sd


def model_init(model_path, device="cuda", max_seq_len=4096, tokenizer=None, **kwargs):
    """videollama2/__init__.py:14-29 for a local checkpoint directory.  Returns (model, processor, tokenizer)."""
    cfg, hf = config_from_checkpoint(model_path)
    check_supported(cfg)
    siglip = cfg["vision"]["family"] == "siglip"
    image_processor = (default_siglip_image_processor if siglip else default_image_processor)(cfg["vision"]["image_size"])
    model = VideoLLaMA2Hip(cfg, load_state_dict(model_path), device, max_seq_len, image_processor=image_processor)
    model.config = types.SimpleNamespace(**hf)
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=True)
    if tokenizer.pad_token is None and tokenizer.unk_token is not None:              # __init__.py:19-20
        tokenizer.pad_token = tokenizer.unk_token
    num_frames = hf.get("num_frames", NUM_FRAMES)
    processor = {"image": partial(process_image, processor=image_processor, aspect_ratio=None),
                 "video": partial(process_video, processor=image_processor, aspect_ratio=None, num_frames=num_frames)}
    return model, processor, tokenizer


def mm_infer(image_or_video, instruct, model, tokenizer, modal="video", **kwargs):
    """videollama2/__init__.py:32-114: tag + chat-template the instruction, put the modal sentinel in the ids, greedy-generate
    with the keyword stopping criterion, decode.  Frames go to the device as the reference sends them (`tensor.half()`,
    __init__.py:60; the patch-row kernel reads fp16 / bf16 / fp32 / uint8 frames alike).  do_sample / temperature / top_p are read
    from kwargs with the reference's defaults (__init__.py:93-95: temperature 0.2 when sampling, top_p 0.9) and handed to generate."""
    if modal == "image":
        modal_token = DEFAULT_IMAGE_TOKEN
    elif modal == "video":
        modal_token = DEFAULT_VIDEO_TOKEN
    elif modal == "text":
        modal_token = ""
    else:
        raise ValueError(f"Unsupported modal: {modal}")
    dev = model.device
    tensor = None if modal == "text" else [((image_or_video if image_or_video.dtype == torch.uint8
                                             else image_or_video.half()).to(dev), modal)]
    if isinstance(instruct, str):
        message = [{"role": "user", "content": modal_token + "\n" + instruct}]
    elif isinstance(instruct, list):
        message = copy.deepcopy(instruct)
        message[0]["content"] = modal_token + "\n" + message[0]["content"]
    else:
        raise ValueError(f"Unsupported type of instruct: {type(instruct)}")
    system = [{"role": "system", "content": _SYSTEM_PROMPT}] if getattr(model.config, "model_type", "") in _SYSTEM_TYPES else []
    prompt = tokenizer.apply_chat_template(system + message, tokenize=False, add_generation_prompt=True)
    input_ids = tokenizer_multimodal_token(prompt, tokenizer, modal_token, return_tensors="pt").unsqueeze(0).long().to(dev)
    attention_masks = input_ids.ne(tokenizer.pad_token_id).long()
    stopping_criteria = KeywordsStoppingCriteria([tokenizer.eos_token], tokenizer, input_ids)
    do_sample = kwargs.get("do_sample", False)
    temperature = kwargs.get("temperature", 0.2 if do_sample else 0.0)         # __init__.py:94 (unused by greedy decoding)
    top_p = kwargs.get("top_p", 0.9)                                           # __init__.py:95
    with torch.inference_mode():
        output_ids = model.generate(input_ids, attention_mask=attention_masks, images=tensor, do_sample=do_sample,
                                    temperature=temperature, top_p=top_p, max_new_tokens=kwargs.get("max_new_tokens", 2048), use_cache=True,
                                    stopping_criteria=[stopping_criteria], pad_token_id=tokenizer.eos_token_id,
                                    eos_token_id=tokenizer.eos_token_id)
    return tokenizer.batch_decode(output_ids, skip_special_tokens=True)[0].strip()
