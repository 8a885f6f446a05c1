"""HipSTCConnector -- drop-in for videollama2/model/projector.py:133-215 `STCConnector` (`mm_projector` built by
build_vision_projector for mm_projector_type == "stc_connector").  forward([b, t, l, d]) -> [b, (t' h' w'), D].
Channels-last ("token-major") activations: 1x1 convs / Conv3d / readout are MFMA GEMMs, LayerNorm2d is a row
LayerNorm, depthwise 3x3 + SE are direct kernels (see csrc/k_stc.h)."""
import torch
import torch.nn as nn

from . import _lib, ops
from .weights import pack_connector


def conv3d_k2s2p1_index(T, H, W, device, to_range=None, frame_lo=0, n_local=None, padding=1):
    """Gather table for Conv3d(kernel 2, stride 2, padding p): p = 1 is STCConnector's sampler (projector.py:164-174), p = 0
    STCConnectorV35's (projector.py:225-238).  Output (to,ho,wo) tap (kt,kh,kw) reads input (2*to-p+kt, 2*ho-p+kh,
    2*wo-p+kw) or zero outside the [T,H,W] volume (with p = 0 nothing falls outside; an odd trailing row/column/frame is
    simply never read, as in nn.Conv3d).
    Sharded form: only the output frames `to_range = (to0, to1)` are produced, from a local row pool that holds the
    `n_local` input frames starting at global frame `frame_lo`.  Returns (int32 [8, n_to*Ho*Wo], (n_to, Ho, Wo))."""
    o = lambda n: (n + 2 * padding - 2) // 2 + 1
    To, Ho, Wo = o(T), o(H), o(W)
    to0, to1 = to_range if to_range is not None else (0, To)
    n_local = T if n_local is None else n_local
    to = torch.arange(to0, to1).view(-1, 1, 1)
    ho = torch.arange(Ho).view(1, Ho, 1)
    wo = torch.arange(Wo).view(1, 1, Wo)
    segs = []
    for kt in range(2):
        for kh in range(2):
            for kw in range(2):
                t, h, w = 2 * to - padding + kt, 2 * ho - padding + kh, 2 * wo - padding + kw
                ok = (t >= 0) & (t < T) & (h >= 0) & (h < H) & (w >= 0) & (w < W)
                tl = t - frame_lo
                if bool((ok & ((tl < 0) | (tl >= n_local))).any()):
                    raise ValueError("conv3d shard: an input frame needed by this output range is not in the local pool")
                idx = (tl * H + h) * W + w
                segs.append(torch.where(ok, idx, torch.full_like(idx, -1)).reshape(-1))
    return torch.stack(segs, 0).to(torch.int32).contiguous().to(device), (to1 - to0, Ho, Wo)


class HipSTCConnector(nn.Module):
    """padding = 1: `stc_connector`; padding = 0: `stc_connector_v35` (projector.py:225-238: the same module with an unpadded
    Conv3d sampler, the VideoLLaMA2.1 checkpoints' projector)."""

    def __init__(self, state_dict, device="cuda", prefix="model.mm_projector.", padding=1):
        super().__init__()
        self.padding = int(padding)
        self._dev = torch.device(device)
        self._elem = _lib.elem()
        self.w = pack_connector(state_dict, self._dev, prefix)
        if self._dev.type == "cuda":
            ops.attach_workspace(self._dev)       # split-K (opt-in, ops.set_splitk): Conv3d taps, s2 on few output frames
        self._idx_cache = {}
        self._stage = None

    def _forward_stage(self, rows, t, hw):
        """One C call for the whole connector (include/vl2hip.h vl2_stc_forward): the same launches as run_s1 / run_sampler /
        run_s2_readout, issued inside libvl2hip.so."""
        if self._stage is None:
            self._stage = ops.stc_desc(self.w)
        key = (t, hw, None, 0, None)
        if key not in self._idx_cache:
            self._idx_cache[key] = conv3d_k2s2p1_index(t, hw, hw, self._dev, padding=self.padding)
        idx, dims = self._idx_cache[key]
        out = torch.empty((dims[0] * dims[1] * dims[2], self.w["ro2_w"].shape[0]), dtype=_lib.elem_dtype(), device=self._dev)
        return ops.stc_forward(self._stage[0], rows, t, hw, idx, dims, out)

    def _bottleneck(self, x, b, F, H, W):
        HW = H * W
        h = ops.gemm(x, b["conv1_w"])
        h = ops.layernorm(h, b["bn1_w"], b["bn1_b"], 1e-5, silu=True)
        if ops.stage_flags() & ops.STAGE_STC_UNFUSED:
            h = ops.dwconv3x3_ln_silu(h, b["dw_w"], b["bn2_w"], b["bn2_b"], F, H, W, 1e-5)
            g = ops.chan_mean(h, F, HW)
            g = ops.small_linear(g, b["fc1_w"], b["fc1_b"], ops.ACT_SILU)
            g = ops.small_linear(g, b["fc2_w"], b["fc2_b"], ops.ACT_SIGMOID)
            ops.se_scale_(h, g, F, HW)
        else:                               # squeeze inside the depthwise kernel, excite + scale as one launch (the chain vl2_stc_forward issues)
            h, g = ops.dwconv3x3_ln_silu_mean(h, b["dw_w"], b["bn2_w"], b["bn2_b"], F, H, W, 1e-5)
            g = ops.small_linear(g, b["fc1_w"], b["fc1_b"], ops.ACT_SILU)
            ops.se_excite_scale_(h, g, b["fc2_w"], b["fc2_b"], F, HW)
        h = ops.gemm(h, b["conv3_w"])
        sc = ops.layernorm(ops.gemm(x, b["ds_w"]), b["dsbn_w"], b["dsbn_b"], 1e-5) if "ds_w" in b else x
        return ops.layernorm(h, b["bn3_w"], b["bn3_b"], 1e-5, res=sc, silu=True)

    # ---- stages (also used one by one by the frame-sharded path in dist.py)
    def run_s1(self, rows, F, hw):
        """rows [F*hw*hw, 1024] bf16 -> s1 output [F*hw*hw, C] (per-frame: shards over frames)."""
        for blk in self.w["s1"]:
            rows = self._bottleneck(rows, blk, F, hw, hw)
        return rows

    def run_sampler(self, pool, T, hw, to_range=None, frame_lo=0, n_local=None):
        """Conv3d(k2,s2,p1)+bias+SiLU as the gathered GEMM over a row pool of s1 frames -> ([n_to*Ho*Wo, C], (n_to,Ho,Wo))."""
        key = (T, hw, to_range, frame_lo, n_local)
        if key not in self._idx_cache:
            self._idx_cache[key] = conv3d_k2s2p1_index(T, hw, hw, self._dev, to_range, frame_lo, n_local, self.padding)
        idx, dims = self._idx_cache[key]
        h = ops.gemm(pool, self.w["samp_w"], bias=self.w["samp_b"], act=ops.ACT_SILU,
                     gather=(idx, self.w["zero_row"], self.w["cin"]))
        return h, dims

    def run_s2_readout(self, h, To, Ho, Wo, return_s2=False):
        """s2 (per output frame) + readout MLP (per token) -> [To*Ho*Wo, D_out]."""
        for blk in self.w["s2"]:
            h = self._bottleneck(h, blk, To, Ho, Wo)
        s2 = h
        h = ops.gemm(h, self.w["ro0_w"], bias=self.w["ro0_b"], act=ops.ACT_GELU)
        h = ops.gemm(h, self.w["ro2_w"], bias=self.w["ro2_b"])
        return (h, s2) if return_s2 else h

    @torch.no_grad()
    def forward(self, x, return_stages=False):
        _lib.check_elem(self._elem, type(self).__name__)
        if x.dim() == 5:                                              # [b, t, h, w, d]  (projector.py:200-201)
            b, t, hh, ww, d = x.shape
            x = x.reshape(b, t, hh * ww, d)
        b, t, l, d = x.shape
        hw = int(l ** 0.5)
        in_dtype = x.dtype
        x = x.to(device=self._dev, dtype=_lib.elem_dtype()).contiguous()
        outs, stages = [], {}
        for bi in range(b):
            if ops.stage_enabled() and not return_stages:
                outs.append(self._forward_stage(x[bi].reshape(t * l, d), t, hw))
                continue
            s1 = self.run_s1(x[bi].reshape(t * l, d), t, hw)
            samp, (To, Ho, Wo) = self.run_sampler(s1, t, hw)
            h, s2 = self.run_s2_readout(samp, To, Ho, Wo, return_s2=True)
            outs.append(h)
            if return_stages:
                stages = dict(s1=s1.view(t, hw, hw, -1), sampler=samp.view(To, Ho, Wo, -1), s2=s2.view(To, Ho, Wo, -1))
        out = (outs[0].unsqueeze(0) if b == 1 else torch.stack(outs, 0)).to(in_dtype)
        return (out, stages) if return_stages else out
