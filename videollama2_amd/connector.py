"""HipSTCConnector -- drop-in for videollama2/model/projector.py:133-215 `STCConnector` (`mm_projector` built by
build_vision_projector for mm_projector_type == "stc_connector").  forward([b, t, l, d]) -> [b, (t' h' w'), D].
Channels-last ("token-major") activations: 1x1 convs / Conv3d / readout are MFMA GEMMs, LayerNorm2d is a row
LayerNorm, depthwise 3x3 + SE are direct kernels (see csrc/k_stc.h)."""
import torch
import torch.nn as nn

from . import ops
from .weights import pack_connector


def conv3d_k2s2p1_index(T, H, W, device):
    """Gather table for Conv3d(kernel 2, stride 2, padding 1) (projector.py:164-174): output (to,ho,wo) tap (kt,kh,kw)
    reads input (2*to-1+kt, 2*ho-1+kh, 2*wo-1+kw) or zero outside.  Returns (int32 [8, To*Ho*Wo], (To,Ho,Wo))."""
    o = lambda n: (n + 2 - 2) // 2 + 1
    To, Ho, Wo = o(T), o(H), o(W)
    to = torch.arange(To).view(To, 1, 1)
    ho = torch.arange(Ho).view(1, Ho, 1)
    wo = torch.arange(Wo).view(1, 1, Wo)
    segs = []
    for kt in range(2):
        for kh in range(2):
            for kw in range(2):
                t, h, w = 2 * to - 1 + kt, 2 * ho - 1 + kh, 2 * wo - 1 + kw
                ok = (t >= 0) & (t < T) & (h >= 0) & (h < H) & (w >= 0) & (w < W)
                idx = (t * H + h) * W + w
                segs.append(torch.where(ok, idx, torch.full_like(idx, -1)).reshape(-1))
    return torch.stack(segs, 0).to(torch.int32).contiguous().to(device), (To, Ho, Wo)


class HipSTCConnector(nn.Module):
    def __init__(self, state_dict, device="cuda", prefix="model.mm_projector."):
        super().__init__()
        self._dev = torch.device(device)
        self.w = pack_connector(state_dict, self._dev, prefix)
        self._idx_cache = {}

    def _bottleneck(self, x, b, F, H, W):
        HW = H * W
        h = ops.gemm(x, b["conv1_w"])
        h = ops.layernorm(h, b["bn1_w"], b["bn1_b"], 1e-5, silu=True)
        h = ops.dwconv3x3_ln_silu(h, b["dw_w"], b["bn2_w"], b["bn2_b"], F, H, W, 1e-5)
        g = ops.chan_mean(h, F, HW)
        g = ops.small_linear(g, b["fc1_w"], b["fc1_b"], ops.ACT_SILU)
        g = ops.small_linear(g, b["fc2_w"], b["fc2_b"], ops.ACT_SIGMOID)
        ops.se_scale_(h, g, F, HW)
        h = ops.gemm(h, b["conv3_w"])
        sc = ops.layernorm(ops.gemm(x, b["ds_w"]), b["dsbn_w"], b["dsbn_b"], 1e-5) if "ds_w" in b else x
        return ops.layernorm(h, b["bn3_w"], b["bn3_b"], 1e-5, res=sc, silu=True)

    @torch.no_grad()
    def forward(self, x, return_stages=False):
        if x.dim() == 5:                                              # [b, t, h, w, d]  (projector.py:200-201)
            b, t, hh, ww, d = x.shape
            x = x.reshape(b, t, hh * ww, d)
        b, t, l, d = x.shape
        hw = int(l ** 0.5)
        in_dtype = x.dtype
        x = x.to(device=self._dev, dtype=torch.bfloat16).contiguous()
        outs, stages = [], {}
        for bi in range(b):
            h = x[bi].reshape(t * l, d)
            for blk in self.w["s1"]:
                h = self._bottleneck(h, blk, t, hw, hw)
            key = (t, hw)
            if key not in self._idx_cache:
                self._idx_cache[key] = conv3d_k2s2p1_index(t, hw, hw, self._dev)
            idx, (To, Ho, Wo) = self._idx_cache[key]
            s1 = h
            h = ops.gemm(h, self.w["samp_w"], bias=self.w["samp_b"], act=ops.ACT_SILU,
                         gather=(idx, self.w["zero_row"], self.w["cin"]))
            samp = h
            for blk in self.w["s2"]:
                h = self._bottleneck(h, blk, To, Ho, Wo)
            s2 = h
            h = ops.gemm(h, self.w["ro0_w"], bias=self.w["ro0_b"], act=ops.ACT_GELU)
            h = ops.gemm(h, self.w["ro2_w"], bias=self.w["ro2_b"])
            outs.append(h)
            if return_stages:
                stages = dict(s1=s1.view(t, hw, hw, -1), sampler=samp.view(To, Ho, Wo, -1), s2=s2.view(To, Ho, Wo, -1))
        out = (outs[0].unsqueeze(0) if b == 1 else torch.stack(outs, 0)).to(in_dtype)
        return (out, stages) if return_stages else out
