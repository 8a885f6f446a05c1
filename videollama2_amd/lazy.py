"""Factory-level drop-ins: what `build_vision_tower(cfg)` (videollama2/model/encoder.py:154-164) and
`build_vision_projector(cfg)` (videollama2/model/projector.py:95-122) return once `install.install()` has patched the two
factories.  Each is an `nn.Module` that hosts its parameters under EXACTLY the reference module's state-dict names (so
`from_pretrained` / `load_state_dict` / `mm_projector.bin` loading of the reference fill them, videollama2_arch.py:96,
model/__init__.py:163-164) but builds no HF / timm module and runs no torch math: the first forward packs the loaded tensors
into the kernel layouts (weights.py) and hands over to HipCLIPVisionTower / HipSiglipVisionTower / HipSTCConnector; the
hosting parameters are released then (`keep_parameters=False`), so the weights live once."""
import types

import torch
import torch.nn as nn

from . import _lib

from .connector import HipSTCConnector
from .tower import (HipCLIPVisionTower, HipSiglipVisionTower, default_image_processor, default_siglip_image_processor)

# public hyper-parameters of the two towers the released checkpoints use (no hub access on the target box)
PUBLIC_TOWERS = {
    "clip-vit-large-patch14-336": dict(family="clip", hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                       num_attention_heads=16, image_size=336, patch_size=14, layer_norm_eps=1e-5),
    "siglip-so400m-patch14-384": dict(family="siglip", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                                      num_attention_heads=16, image_size=384, patch_size=14, layer_norm_eps=1e-6),
}


class ParamHost(nn.Module):
    """Parameters registered under dotted names as a tree of empty Modules: `host.state_dict()` has exactly those keys."""

    def __init__(self, names_shapes, dtype=None, device="cpu"):
        super().__init__()
        dtype = _lib.elem_dtype() if dtype is None else dtype
        for name, shape in names_shapes:
            mod, parts = self, name.split(".")
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            mod.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=dtype, device=device), requires_grad=False))


def tower_config(name, select_layer=-2):
    """Vision hyper-parameters from a local tower directory (config.json) or the public values keyed by the tower's name."""
    import json
    import os
    need = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size", "patch_size")
    v, fam = None, None
    if os.path.isfile(os.path.join(str(name), "config.json")):
        top = json.load(open(os.path.join(str(name), "config.json")))
        # the standard tower directories (openai/clip-vit-large-patch14-336, google/siglip-so400m-patch14-384) ship a CLIPConfig /
        # SiglipConfig whose vision hyper-parameters sit under `vision_config`; CLIPVisionConfig.from_pretrained reads that level
        v = top.get("vision_config", top)
        fam = "siglip" if "siglip" in (str(top.get("model_type", "")) + str(v.get("model_type", "")) + str(name)).lower() else "clip"
        if not all(k in v for k in need):
            v = None                                                                   # partial config: fall back to the public values
    if v is None:
        key = str(name).strip("/").split("/")[-1]
        if key not in PUBLIC_TOWERS:
            raise ValueError(f"Unknown vision tower: {name}")                          # encoder.py:162
        v = dict(PUBLIC_TOWERS[key])
        fam = v["family"]
    return dict(family=fam, hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"], num_hidden_layers=v["num_hidden_layers"],
                num_attention_heads=v["num_attention_heads"], image_size=v["image_size"], patch_size=v["patch_size"],
                layer_norm_eps=v.get("layer_norm_eps", 1e-5), select_layer=select_layer)


def default_key_layout():
    """Key layout of HF CLIPVisionModel / SiglipVisionModel in the INSTALLED transformers: releases before 5.0 (the reference pins
    4.40.0, and the released checkpoints were written by it) keep the encoder under a `vision_model.` level, 5.x does not.  The
    hosting module must use the layout `from_pretrained(low_cpu_mem_usage=True)` will match by name (load_state_dict hooks do not
    run on that path)."""
    try:
        import transformers
        return "vision_model" if int(transformers.__version__.split(".")[0]) < 5 else "flat"
    except Exception:
        return "flat"


def _tower_names(v):
    """State-dict names of HF CLIPVisionModel / SiglipVisionModel below the reference tower's `vision_tower` attribute
    (transformers 5.x naming, no `vision_model.` level; LazyHipVisionTower adds the level for the 4.x layout)."""
    from .weights import state_dict_names
    cfg = dict(vision=v, llm=dict(hidden_size=128, intermediate_size=128, num_hidden_layers=0, num_attention_heads=1,
                                  num_key_value_heads=1, head_dim=128, vocab_size=128))
    pre = "model.vision_tower."
    out = [(n[len(pre):], s) for n, s in state_dict_names(cfg) if n.startswith(pre)]
    D, I = v["hidden_size"], v["intermediate_size"]
    # present in the HF modules (and in real checkpoints), never reached by hidden_states[-2]: hosted so that a strict load works
    out += [("vision_tower.post_layernorm.weight", (D,)), ("vision_tower.post_layernorm.bias", (D,))]
    if v.get("family", "clip") == "siglip":        # SiglipMultiheadAttentionPoolingHead (HF:models/siglip/modeling_siglip.py)
        h = "vision_tower.head."
        out += [(h + "probe", (1, 1, D)), (h + "attention.in_proj_weight", (3 * D, D)), (h + "attention.in_proj_bias", (3 * D,)),
                (h + "attention.out_proj.weight", (D, D)), (h + "attention.out_proj.bias", (D,)), (h + "layernorm.weight", (D,)),
                (h + "layernorm.bias", (D,)), (h + "mlp.fc1.weight", (I, D)), (h + "mlp.fc1.bias", (I,)),
                (h + "mlp.fc2.weight", (D, I)), (h + "mlp.fc2.bias", (D,))]
    return out


class LazyHipVisionTower(nn.Module):
    """What the patched `build_vision_tower` returns (reference: CLIPVisionTower / SiglipVisionTower, encoder.py:12-151)."""

    def __init__(self, vision_tower, args, device=None, keep_parameters=False, key_layout="auto", **_unused):
        """key_layout: "vision_model" = transformers 4.x names (`vision_tower.vision_model.encoder...`, the released checkpoints),
        "flat" = transformers 5.x names, "auto" = whatever the installed transformers uses (`default_key_layout`).  A plain
        `load_state_dict` accepts either layout (pre-hook); `from_pretrained(low_cpu_mem_usage=True)` matches by name only."""
        super().__init__()
        self.vision_tower_name = vision_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        v = tower_config(vision_tower, self.select_layer)
        self._v = v
        self._hip = []                                          # [HipCLIPVisionTower] once packed (a list: not a submodule)
        self._device = device
        self._keep = keep_parameters
        self.is_loaded = True
        try:                                                    # encoder.py:21 / :94 (needs the tower's files; public values otherwise)
            from transformers import CLIPImageProcessor, SiglipImageProcessor
            cls = SiglipImageProcessor if v["family"] == "siglip" else CLIPImageProcessor
            self.image_processor = cls.from_pretrained(vision_tower)
        except Exception:
            self.image_processor = (default_siglip_image_processor if v["family"] == "siglip" else default_image_processor)(v["image_size"])
        self.key_layout = default_key_layout() if key_layout == "auto" else key_layout
        if self.key_layout not in ("vision_model", "flat"):
            raise ValueError(f"key_layout {key_layout!r}: expected 'auto', 'vision_model' or 'flat'")
        names = _tower_names(v)
        if self.key_layout == "vision_model":
            names = [("vision_tower.vision_model." + n[len("vision_tower."):], s) for n, s in names]
        host = ParamHost(names)
        self.vision_tower = host._modules["vision_tower"]       # same attribute name as the reference -> same state-dict keys
        self.config = types.SimpleNamespace(**v)
        self._register_load_state_dict_pre_hook(self._rename_other_layout)

    def _rename_other_layout(self, state_dict, prefix, *_):
        """`load_state_dict` of a checkpoint written in the OTHER key layout (4.x <-> 5.x): rename its keys to the hosted ones.
        Also records which tensors arrived through `load_state_dict` (check_loaded trusts those: a synthetic or freshly initialised
        tower with constant LayerNorm scales is a legitimate load)."""
        norm = lambda k: k.split("vision_model.")[-1].split("vision_tower.")[-1]
        hosted = {norm(k) for k in self.vision_tower.state_dict()}       # only keys that land in a hosted parameter count as loaded
        self._seen_in_load = getattr(self, "_seen_in_load", set()) | ({norm(k[len(prefix):]) for k in state_dict if k.startswith(prefix + "vision_tower.")} & hosted)
        old = prefix + "vision_tower.vision_model."
        if self.key_layout == "flat":
            for k in [k for k in state_dict if k.startswith(old)]:
                state_dict[prefix + "vision_tower." + k[len(old):]] = state_dict.pop(k)
        elif not any(k.startswith(old) for k in state_dict):
            for k in [k for k in state_dict if k.startswith(prefix + "vision_tower.")]:
                state_dict[old + k[len(prefix + "vision_tower."):]] = state_dict.pop(k)

    def to_empty(self, *args, **kwargs):
        """Re-materialising the parameters (meta -> device storage) voids what earlier `load_state_dict` calls put there: the loaded-by-construction
        record starts over, so a tower re-created after a load is judged by its values again."""
        self._seen_in_load = set()
        return super().to_empty(*args, **kwargs)

    def check_loaded(self):
        """Refuse to run on weights that were never loaded.  `from_pretrained(low_cpu_mem_usage=True)` matches checkpoint keys to
        parameter names and materialises every UNMATCHED parameter with uninitialised memory (a warning, not an error); a key-layout
        mismatch would therefore run the tower on garbage.  Signature of uninitialised / never-written storage: non-finite values,
        or a weight matrix / LayerNorm scale that is constant (fresh pages are zero).  A trained tower has neither."""
        bad = []
        seen = getattr(self, "_seen_in_load", set())                 # tensors that came through load_state_dict: loaded by construction
        for k, p in self.vision_tower.state_dict().items():
            if p.device.type == "meta":
                bad.append(k + " (meta)")
                continue
            if "post_layernorm" in k or ".head." in k:               # hosted for strict loading only, never read
                continue
            t = p.detach().float()
            if not bool(torch.isfinite(t).all()):
                bad.append(k + " (non-finite)")
            # the value heuristic is for the by-name loaders that run no hook (from_pretrained(low_cpu_mem_usage=True)): only LARGE
            # weight matrices are judged (a constant LayerNorm scale or a tiny test matrix is not evidence), `strict_loaded_check =
            # False` on the instance switches it off
            elif getattr(self, "strict_loaded_check", True) and k.split("vision_model.")[-1] not in seen \
                    and t.dim() >= 2 and t.numel() >= 4096 and float(t.max() - t.min()) == 0.0:
                bad.append(k + " (constant)")
        if bad:
            raise RuntimeError(f"vision tower parameters were never loaded ({len(bad)} tensors, e.g. {bad[:4]}): the checkpoint's key "
                               f"layout does not match the hosted one ('{self.key_layout}'); build the tower with "
                               f"key_layout='{'flat' if self.key_layout == 'vision_model' else 'vision_model'}' or load with load_state_dict")

    def pack(self, device=None):
        if not self._hip:
            dev = torch.device(device or self._device or next(self.vision_tower.parameters()).device)
            if dev.type == "meta":
                raise RuntimeError("vision tower parameters were never loaded (still on the meta device)")
            self.check_loaded()
            cfg = dict(vision=self._v)
            cls = HipSiglipVisionTower if self._v["family"] == "siglip" else HipCLIPVisionTower
            strip = "vision_model." if self.key_layout == "vision_model" else ""
            sd = {"vision_tower." + (k[len(strip):] if k.startswith(strip) else k): p for k, p in self.vision_tower.state_dict().items()}
            self._hip.append(cls(cfg, sd, dev, select_feature=self.select_feature, image_processor=self.image_processor, prefix="vision_tower."))
            if not self._keep:
                self.vision_tower = None                         # the packed copies are the weights now
        return self._hip[0]

    @torch.no_grad()
    def forward(self, images):
        return self.pack(images[0].device if type(images) is list else images.device)(images)

    # ---- attributes the reference reads (encoder.py:55-81)
    dtype = property(lambda self: _lib.elem_dtype())
    device = property(lambda self: self._hip[0].device if self._hip else next(self.vision_tower.parameters()).device)
    hidden_size = property(lambda self: self._v["hidden_size"])
    num_patches_per_side = property(lambda self: self._v["image_size"] // self._v["patch_size"])
    num_patches = property(lambda self: self.num_patches_per_side ** 2)
    image_size = property(lambda self: self._v["image_size"])


def _connector_names(cin, D):
    from .weights import state_dict_names
    cfg = dict(vision=dict(hidden_size=cin, intermediate_size=128, num_hidden_layers=0, num_attention_heads=1, image_size=28, patch_size=14),
               llm=dict(hidden_size=D, intermediate_size=128, num_hidden_layers=0, num_attention_heads=1, num_key_value_heads=1,
                        head_dim=128, vocab_size=128))
    pre = "model.mm_projector."
    return [(n[len(pre):], s) for n, s in state_dict_names(cfg) if n.startswith(pre)]


class LazyHipSTCConnector(nn.Module):
    """What the patched `build_vision_projector` returns for mm_projector_type stc_connector / stc_connector_v35 (reference:
    STCConnector / STCConnectorV35, projector.py:133-238): parameters under timm RegStage's names, no timm import."""

    def __init__(self, config, device=None, keep_parameters=False):
        super().__init__()
        ptype = getattr(config, "mm_projector_type", "stc_connector")
        if ptype not in ("stc_connector", "stc_connector_v35"):
            raise ValueError(f"Unknown projector type: {ptype}")                      # projector.py:122
        self.padding = 0 if ptype == "stc_connector_v35" else 1
        self._hip, self._device, self._keep = [], device, keep_parameters
        host = ParamHost(_connector_names(config.mm_hidden_size, config.hidden_size))
        for name, mod in host._modules.items():                   # s1, sampler, s2, readout at the top level, like the reference
            self.add_module(name, mod)

    def pack(self, device=None):
        if not self._hip:
            dev = torch.device(device or self._device or next(self.parameters()).device)
            if dev.type == "meta":
                raise RuntimeError("mm_projector parameters were never loaded (still on the meta device)")
            self._hip.append(HipSTCConnector(dict(self.state_dict()), dev, prefix="", padding=self.padding))
            if not self._keep:
                for name in list(self._modules):
                    del self._modules[name]
        return self._hip[0]

    @torch.no_grad()
    def forward(self, x, *args, **kwargs):
        return self.pack(x.device)(x, *args, **kwargs)
