"""HipCLIPVisionTower -- drop-in for videollama2/model/encoder.py:12-81 `CLIPVisionTower` (the object
`build_vision_tower(cfg)` returns, encoder.py:154-164): same call contract `tower(frames[(b t),3,H,W]) ->
[(b t), num_patches, hidden]` in the input dtype, same attributes (`hidden_size`, `num_patches`,
`num_patches_per_side`, `image_size`, `dtype`, `device`, `config`, `image_processor`), `@torch.no_grad()`.
All arithmetic runs in libvl2hip.so (gfx950); torch only owns the buffers."""
import types

import torch
import torch.nn as nn

from . import _lib, ops
from .weights import pack_siglip_tower, pack_tower


def default_image_processor(image_size=336):
    """The CLIPImageProcessor the reference gets from `CLIPImageProcessor.from_pretrained(tower)` (encoder.py:21), built
    from the public openai/clip-vit-large-patch14-336 preprocessor values (no hub access): shortest-edge resize
    (bicubic) -> centre crop -> 1/255 -> mean/std normalise."""
    from transformers import CLIPImageProcessor
    return CLIPImageProcessor(do_resize=True, size={"shortest_edge": image_size}, resample=3, do_center_crop=True,
                              crop_size={"height": image_size, "width": image_size}, do_rescale=True,
                              rescale_factor=1 / 255, do_normalize=True, do_convert_rgb=True,
                              image_mean=[0.48145466, 0.4578275, 0.40821073],
                              image_std=[0.26862954, 0.26130258, 0.27577711])


class HipCLIPVisionTower(nn.Module):
    def __init__(self, cfg, state_dict, device="cuda", select_feature="patch", image_processor=None,
                 prefix="model.vision_tower.vision_tower."):
        super().__init__()
        v = cfg["vision"]
        self.cfg = cfg
        self.select_layer = v["select_layer"]
        self.select_feature = select_feature
        if select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {select_feature}")      # encoder.py:38
        self.image_processor = image_processor
        self._dev = torch.device(device)
        self._elem = _lib.elem()            # the element type the weights are packed in (checked on every call: _lib.check_elem)
        self.w = self._pack(state_dict, cfg, self._dev, prefix)
        if self._dev.type == "cuda":
            ops.attach_workspace(self._dev)       # split-K (opt-in, ops.set_splitk) for small per-rank grids
        self.config = types.SimpleNamespace(**v)
        self.is_loaded = True
        # frames as chunks on this many HIP streams.  Measured on MI355X (encode ms, 1 / 2 / 3 / 4 streams): T=16 14.15 /
        # 14.01 / 13.76-13.88 / 16.15, T=8 8.70 / - / 8.85, T=32 25.25 / - / 25.36: within noise of nothing (the chip is
        # power-limited, idle CUs in a GEMM's last round already return their power to the busy ones) -> default 1.
        self.streams = 1
        self._side = None
        self._stage = None                  # (vl2_vit_desc, keepalive): the whole tower as ONE C call (ops.vit_forward)

    _pack = staticmethod(pack_tower)
    _cls_tokens = 1                     # rows per frame in front of the patches (CLIP: the CLS token)
    _stage_family, _stage_act = 0, ops.ACT_QGELU

    def _hidden_stage(self, images, T, u8, out):
        """The stage-level entry point (include/vl2hip.h vl2_vit_forward): patch embedding + every encoder layer in one call into
        libvl2hip.so -- the same kernels in the same order as `_hidden` below (asserted bit-identical in the tests)."""
        v = self.cfg["vision"]
        if self._stage is None:
            self._stage = ops.vit_desc(self.w, v, self._stage_family, self._stage_act)
        N1 = (v["image_size"] // v["patch_size"]) ** 2 + self._cls_tokens
        if out is None:
            out = torch.empty((T * N1, v["hidden_size"]), dtype=_lib.elem_dtype(), device=self._dev)
        nrm = None
        if u8:
            ip = self.image_processor
            rescale, mean, std = self._default_norm
            if ip is not None:
                rescale, mean, std = getattr(ip, "rescale_factor", rescale), getattr(ip, "image_mean", mean), getattr(ip, "image_std", std)
            nrm = [rescale, *mean, *std]
        ops.vit_forward(self._stage[0], images.contiguous(), T, out, nrm)
        return out, T, N1

    _default_norm = (1 / 255, (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))   # openai/clip preprocessor

    @staticmethod
    def _frame_geometry(images):
        """Float frames [T,3,H,W] (what `process_video` returns) or raw uint8 frames [T,H,W,3] (`mm_utils.process_video_u8`)."""
        if images.dim() == 4 and images.dtype == torch.uint8 and images.shape[-1] == 3:
            return images.shape[0], images.shape[1], images.shape[2], True
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"expected frames [T,3,H,W] (or uint8 [T,H,W,3]), got {tuple(images.shape)}")
        return images.shape[0], images.shape[2], images.shape[3], False

    def _patch_rows(self, images, u8, P, kp):
        if not u8:
            return ops.patchify(images, P, kp)
        ip = self.image_processor               # the arithmetic tail of the processor, applied in registers
        rescale, mean, std = self._default_norm
        if ip is not None:
            rescale, mean, std = getattr(ip, "rescale_factor", rescale), getattr(ip, "image_mean", mean), getattr(ip, "image_std", std)
        return ops.patchify_u8(images, P, kp, rescale, mean, std)

    # ---- attributes the reference reads (encoder.py:55-81, videollama2_arch.py:66, model/__init__.py:186)
    @property
    def dtype(self):
        return _lib.elem_dtype()

    @property
    def device(self):
        return self._dev

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def num_patches(self):
        return self.num_patches_per_side ** 2

    @property
    def image_size(self):
        return self.config.image_size

    @torch.no_grad()
    def forward_hidden(self, images):
        """Returns hidden_states[select_layer] INCLUDING the CLS row: bf16 [T*(N+1), D] (flat token-major).
        With `self.streams` > 1 the (independent) frames run as that many interleaved chunks on separate HIP streams,
        so one chunk's partial last round of GEMM tiles overlaps the other chunk's kernels."""
        _lib.check_elem(self._elem, type(self).__name__)
        ns = min(int(self.streams), images.shape[0] // 4 if images.dim() == 4 else 1)    # >= 4 frames per chunk
        if images.dtype == torch.uint8:
            ns = 1
        if ns <= 1:
            return self._hidden(images)
        v = self.cfg["vision"]
        T = images.shape[0]
        N1 = (images.shape[2] // v["patch_size"]) ** 2 + self._cls_tokens
        images = images.to(self._dev)
        out = torch.empty((T * N1, v["hidden_size"]), dtype=_lib.elem_dtype(), device=self._dev)
        cur = torch.cuda.current_stream(self._dev)
        if self._side is None or len(self._side) != ns:
            self._side = [torch.cuda.Stream(self._dev) for _ in range(ns)]
        bounds = [T * i // ns for i in range(ns + 1)]
        self._multi_stream = True            # (the per-operator loops then keep off the persistent GEMM: its tile counters live in ONE shared workspace)
        try:
            for i, st in enumerate(self._side):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    self._hidden(images[bounds[i]:bounds[i + 1]], out[bounds[i] * N1:bounds[i + 1] * N1])
        finally:
            self._multi_stream = False
        for st in self._side:
            cur.wait_stream(st)
        return out, T, N1

    def _hidden(self, images, out=None):
        v = self.cfg["vision"]
        T, H, W, u8 = self._frame_geometry(images)
        if H != v["image_size"] or W != v["image_size"]:                          # HF:modeling_clip.py:203-207
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({v['image_size']}*{v['image_size']}).")
        images = images.to(self._dev)
        if ops.stage_enabled() and images.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.uint8):
            return self._hidden_stage(images, T, u8, out)
        if getattr(self, "_multi_stream", False):
            return self._hidden_ops(images, T, H, u8, out)
        with ops.tower_gemm_flags():                  # the same GEMM forms as the stage call takes (persistent q/k/v and fc1)
            return self._hidden_ops(images, T, H, u8, out)

    def _hidden_ops(self, images, T, H, u8, out):
        v = self.cfg["vision"]
        w = self.w
        D, P = v["hidden_size"], v["patch_size"]
        G = H // P
        N1 = G * G + 1
        nh = v["num_attention_heads"]
        hd = D // nh
        eps = v["layer_norm_eps"]
        a = self._patch_rows(images, u8, P, w["kp"])
        x = torch.empty((T * N1, D), dtype=_lib.elem_dtype(), device=self._dev)
        ops.gemm(a, w["patch_w"], res=w["pos"], out=x, out_map=(G * G, 1, 1), res_map=(G * G, 1), flop_k=3 * P * P)
        ops.fill_cls(x, w["cls_pos"], T, N1)
        x = ops.layernorm(x, w["pre_w"], w["pre_b"], eps)
        o = torch.empty((T * N1, D), dtype=_lib.elem_dtype(), device=self._dev)
        # layer_norm1 / layer_norm2 never run as kernels: every GEMM that writes the residual stream also emits its row
        # statistics (`stats_out`), and the q/k/v and fc1 GEMMs normalise in their epilogue (weights.fold_norm, csrc/k_gemm.h)
        # The K/64 partial statistics of a row are reduced ONCE per tensor (`row_norm_finalize`, a 2 us kernel) instead of in every
        # column tile of the consuming GEMM (measured, scripts/gemm_norm_bench.py: fc1 +11.4 us per call with the in-tile reduction,
        # +4.5 us with the reduced form).
        rs = ops.row_stats(x)                                                      # seeds the chain (pre_layrnorm's output)
        rn = ops.row_norm_finalize(rs, D, ops.NORM_LN, eps)
        last = len(w["layers"]) - 1
        for li, lw in enumerate(w["layers"]):
            qkv = ops.gemm(x, lw["wqkv"], bias=lw["bqkv"], norm=(ops.NORM_LN, rn, eps, lw["sqkv"]))   # [T*N1, 3D] = q | k | v
            st = (N1 * 3 * D, hd, 3 * D)
            ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, st, st, st, (N1 * D, hd, D), T, nh, N1, N1, 1,
                         hd ** -0.5, False, 0, hd)
            x = ops.gemm(o, lw["wo"], bias=lw["bo"], res=x, stats_out=rs)
            ops.row_norm_finalize(rs, D, ops.NORM_LN, eps, out=rn)
            h = ops.gemm(x, lw["w1"], bias=lw["b1"], act=ops.ACT_QGELU, norm=(ops.NORM_LN, rn, eps, lw["s1"]))
            x = ops.gemm(h, lw["w2"], bias=lw["b2"], res=x, out=out if li == last else None, stats_out=None if li == last else rs)
            if li != last:
                ops.row_norm_finalize(rs, D, ops.NORM_LN, eps, out=rn)
        return x, T, N1

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:                                                   # encoder.py:43-48
            return [self.forward(im.unsqueeze(0)) for im in images]
        in_dtype = images.dtype if images.dtype != torch.uint8 else _lib.elem_dtype()
        x, T, N1 = self.forward_hidden(images)
        x = x.view(T, N1, -1)
        if self.select_feature == "patch":                                         # encoder.py:33-34
            x = x[:, 1:]
        return x.contiguous().to(in_dtype)                                         # encoder.py:51 `.to(images.dtype)`


def default_siglip_image_processor(image_size=384):
    """The SiglipImageProcessor the reference gets from `SiglipImageProcessor.from_pretrained(tower)` (encoder.py:94), built
    from the public google/siglip-so400m-patch14-384 preprocessor values (no hub access): plain bicubic resize to
    image_size^2 -> 1/255 -> (x - 0.5) / 0.5."""
    from transformers import SiglipImageProcessor
    return SiglipImageProcessor(do_resize=True, size={"height": image_size, "width": image_size}, resample=3, do_rescale=True,
                                rescale_factor=1 / 255, do_normalize=True, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])


class HipSiglipVisionTower(HipCLIPVisionTower):
    """Drop-in for videollama2/model/encoder.py:84-151 `SiglipVisionTower` (what `build_vision_tower` returns when the tower
    name contains 'siglip', encoder.py:159-160): `tower(frames[(b t),3,H,W]) -> [(b t), num_patches, hidden]` =
    hidden_states[select_layer] of HF SiglipVisionModel with NO token dropped (feature_select 'patch' is the identity there,
    encoder.py:103-109).  Differences from the CLIP tower, all in HF:models/siglip/modeling_siglip.py: biased patch conv, no
    CLS / no pre-LayerNorm, gelu_pytorch_tanh MLP, and shapes (head_dim 72, MLP 4304) that are zero-padded at load time
    (weights.pack_siglip_tower)."""
    _pack = staticmethod(pack_siglip_tower)
    _cls_tokens = 0
    _stage_family, _stage_act = 1, ops.ACT_GELU_TANH
    _default_norm = (1 / 255, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))                  # google/siglip-so400m-patch14-384 preprocessor

    def __init__(self, cfg, state_dict, device="cuda", select_feature="patch", image_processor=None,
                 prefix="model.vision_tower.vision_tower."):
        if select_feature != "patch":
            raise ValueError(f"Unexpected select feature: {select_feature}")      # encoder.py:108
        super().__init__(cfg, state_dict, device, select_feature, image_processor, prefix)

    def _hidden(self, images, out=None):
        v = self.cfg["vision"]
        T, H, W, u8 = self._frame_geometry(images)
        if H != v["image_size"] or W != v["image_size"]:                          # HF:modeling_siglip.py SiglipVisionEmbeddings
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({v['image_size']}*{v['image_size']}).")
        images = images.to(self._dev)
        if ops.stage_enabled() and images.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.uint8):
            return self._hidden_stage(images, T, u8, out)
        if getattr(self, "_multi_stream", False):
            return self._hidden_ops(images, T, H, u8, out)
        with ops.tower_gemm_flags():
            return self._hidden_ops(images, T, H, u8, out)

    def _hidden_ops(self, images, T, H, u8, out):
        v = self.cfg["vision"]
        w = self.w
        D, P, nh = v["hidden_size"], v["patch_size"], v["num_attention_heads"]
        G = H // P
        N = G * G
        hdp, eps = w["hdp"], v["layer_norm_eps"]
        Hh = nh * hdp
        a = self._patch_rows(images, u8, P, w["kp"])
        x = torch.empty((T * N, D), dtype=_lib.elem_dtype(), device=self._dev)
        ops.gemm(a, w["patch_w"], bias=w["patch_b"], res=w["pos"], out=x, res_map=(N, 0), flop_k=3 * P * P)
        o = torch.empty((T * N, Hh), dtype=_lib.elem_dtype(), device=self._dev)
        rs = ops.row_stats(x)                                                      # norm-carrying chain, as in the CLIP tower
        rn = ops.row_norm_finalize(rs, D, ops.NORM_LN, eps)
        last = len(w["layers"]) - 1
        for li, lw in enumerate(w["layers"]):
            qkv = ops.gemm(x, lw["wqkv"], bias=lw["bqkv"], norm=(ops.NORM_LN, rn, eps, lw["sqkv"]))   # q | k | v, heads padded to hdp
            st = (N * 3 * Hh, hdp, 3 * Hh)
            ops.attn_fwd(qkv, qkv[:, Hh:], qkv[:, 2 * Hh:], o, st, st, st, (N * Hh, hdp, Hh), T, nh, N, N, 1,
                         w["hd"] ** -0.5, False, 0, hdp)
            x = ops.gemm(o, lw["wo"], bias=lw["bo"], res=x, stats_out=rs)
            ops.row_norm_finalize(rs, D, ops.NORM_LN, eps, out=rn)
            h = ops.gemm(x, lw["w1"], bias=lw["b1"], act=ops.ACT_GELU_TANH, norm=(ops.NORM_LN, rn, eps, lw["s1"]))
            x = ops.gemm(h, lw["w2"], bias=lw["b2"], res=x, out=out if li == last else None, stats_out=None if li == last else rs)
            if li != last:
                ops.row_norm_finalize(rs, D, ops.NORM_LN, eps, out=rn)
        return x, T, N

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:                                                   # encoder.py:113-118
            return [self.forward(im.unsqueeze(0)) for im in images]
        x, T, N = self.forward_hidden(images)
        return x.view(T, N, -1).to(images.dtype if images.dtype != torch.uint8 else _lib.elem_dtype())   # encoder.py:121
