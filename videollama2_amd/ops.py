"""Operator-level Python wrappers over the C ABI (include/vl2hip.h).  PyTorch-ROCm tensors are used ONLY as device
storage: every wrapper passes raw `data_ptr()`s plus the current HIP stream to libvl2hip.so and returns torch tensors
it allocated.  No torch math happens here and there is no fallback path."""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_QGELU, ACT_GELU, ACT_SILU, ACT_SIGMOID, ACT_GELU_TANH = 0, 1, 2, 3, 4, 5
GEMM_SWIGLU, GEMM_OUT_F32 = 1, 2
PROFILE = None      # bench.py sets this to a list: every gemm launch is then bracketed by HIP events on the launch stream


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.dim() >= 2 and t.stride(-1) != 1:
        raise ValueError(f"{name} must be contiguous in its last dim")


GEMM_SPLITK = 4
NORM_NONE, NORM_RMS, NORM_LN = 0, 1, 2

# Launch controls of the GEMM family.  They are HOST-side settings of this binding layer, handed to libvl2hip.so with every
# call (vl2_gemm_desc.variant / flags / ws): the library itself holds no mutable state.
_WORKSPACE = {}
_CTL = dict(variant=0, splitk=False, attn_variant=0, stage_flags=0, gemm_flags=0)
GEMM_PERSISTENT, GEMM_NO_MIX, GEMM_NO_FILL, GEMM_WEAVE, GEMM_FP8 = 8, 16, 32, 64, 128                                    # vl2_gemm_desc.flags (include/vl2hip.h)
STAGE_PERSISTENT_GEMM, STAGE_NO_MIX, STAGE_SELF_REDUCE, STAGE_FUSED_DECODE_ATTN, STAGE_DECODE_TAIL, STAGE_STC_UNFUSED = 1, 2, 4, 8, 16, 32
STAGE_WEAVE = 256                     # lab: LDS-DMA issue woven into the MFMA phases of the 128x256 / 224x128 / 192x128 ping-pong GEMMs
STAGE_NO_FILL_TILES = 128             # A/B of the fill-the-round GEMM tiles (csrc/k_gemm7.h)
STAGE_NO_WEAVE4, GEMM_NO_WEAVE4 = 16384, 1024   # A/B: the 192-row tiles without the woven issue (their default since round 5)
STAGE_WEAVE4, GEMM_WEAVE4 = 8192, 512 # the 256x256 / 192x256 ping-pong GEMMs issue their LDS-DMA from the matrix phases (k_gemm.h gemm4_body WEAVE4)
STAGE_NO_MFMA16 = 65536               # the decoder prefill's gate/up WITHOUT the 16x16x32-MFMA kernels (its default since round 6: decoder.prefill passes mfma16=True)
STAGE_MFMA16, GEMM_MFMA16 = 32768, 2048  # opt-in: every SwiGLU (gate/up) GEMM of the session on the 16x16x32-MFMA kernel (csrc/k_gemm9.h): as accurate, OTHER last bits
STAGE_VIT_NO_PERSISTENT = 4096        # vl2_vit_forward without the persistent GEMM form (its default since round 5): A/B
STAGE_NO_TICKET_OPS = 2048            # ops.gemm(norm_out=...) only: the appended launch instead of the in-kernel ticket (test / A/B control of the operator path)
STAGE_ROW_TICKET, GEMM_NO_TICKET = 1024, 256  # stage calls: producer-side finalize (k_gemm.h gemm_rows_ticket) instead of the row_norm_finalize launches (lab: not faster)
STAGE_PREFILL_FP8 = 512               # prefill projections on the fp8 matrix pipe (W8A8, vl2_llm_desc.layers_w8); set by the decoder, not a lab switch
STAGE_DECODE_FP8 = 64                 # decode step on the fp8 copies of the weights (vl2_llm_desc.layers_w8); set by the decoder, not a lab switch
GEMV_RMS_PLAIN = 32   # vl2_*_desc.flags of the stage calls


def attach_workspace(device):
    """Allocate (once per device, zero-filled) the GEMM workspace (split-K partial tiles + tile counters, stream-K, the skinny-M
    GEMM's fp32 partial sums); the library itself never allocates.  Calls on one device share it, so they must be ordered on
    one stream (every path in this package runs its GEMMs on the current stream)."""
    key = str(torch.device(device))
    if key not in _WORKSPACE:
        n = int(_lib.load().vl2_workspace_bytes())
        _WORKSPACE[key] = torch.zeros((n + 15) // 16 * 16, dtype=torch.uint8, device=device)
    return _WORKSPACE[key]


def _ws(device):
    return _WORKSPACE.get(str(torch.device(device)))


def set_splitk(on):
    """Split-K for small-grid GEMMs (VL2_GEMM_SPLITK; needs `attach_workspace`).  Off by default: it trades the "same rows ->
    same bits whatever M" property for latency on small-M shapes (measured on MI355X: 1154x1024x4096 45.9 -> 39.7 us,
    169x4096x4096 40.8 -> 27.9 us, 338x4096x32768 436 -> 160 us).  Never enable it while GEMMs run concurrently on several
    streams (they would share the one workspace)."""
    _CTL["splitk"] = bool(on)


def set_attn_kv_groups(n):
    """Attention structure (vl2_attn_fwd `variant`, include/vl2hip.h): 0 = auto, 1 = register-staged with one group of 4 waves per workgroup,
    2 = two groups that split the KV tiles and merge through LDS, 3 = LDS-DMA ring + transpose reads, 4 = 3 with two key streams."""
    _CTL["attn_variant"] = int(n)


def set_stage_flags(flags):
    """Experiment controls of the stage-level entry points and of `gemm` (VL2_STAGE_* in include/vl2hip.h; all off by default):
    STAGE_PERSISTENT_GEMM, STAGE_NO_MIX, STAGE_SELF_REDUCE, STAGE_FUSED_DECODE_ATTN.  They travel in the descriptors of the calls
    (the library reads no environment variables and keeps no state)."""
    _CTL["stage_flags"] = int(flags)
    _CTL["gemm_flags"] = (GEMM_PERSISTENT if flags & STAGE_PERSISTENT_GEMM else 0) | (GEMM_NO_MIX if flags & STAGE_NO_MIX else 0) | \
                         (GEMM_NO_FILL if flags & STAGE_NO_FILL_TILES else 0) | (GEMM_WEAVE if flags & STAGE_WEAVE else 0) | \
                         (GEMM_NO_TICKET if flags & STAGE_NO_TICKET_OPS else 0) | (GEMM_WEAVE4 if flags & STAGE_WEAVE4 else 0) | \
                         (GEMM_NO_WEAVE4 if flags & STAGE_NO_WEAVE4 else 0)


def stage_flags():
    return _CTL["stage_flags"]


class tower_gemm_flags:
    """The per-operator tower loops (tower.py `_hidden`) mirror what vl2_vit_forward does with its GEMMs: the persistent form (VL2_GEMM_PERSISTENT) is
    the tower's default since round 5, unless STAGE_VIT_NO_PERSISTENT is set.  A context manager around those loops; restores the session's flags."""
    def __enter__(self):
        self._saved = _CTL["gemm_flags"]
        if not (_CTL["stage_flags"] & STAGE_VIT_NO_PERSISTENT):
            _CTL["gemm_flags"] |= GEMM_PERSISTENT
        return self

    def __exit__(self, *exc):
        _CTL["gemm_flags"] = self._saved
        return False


def set_gemm_variant(v):
    """0 auto (per-shape choice), 1 128x128x64, 2 stream-K, 4 128x256x64 ping-pong, 8 256x256x32 ping-pong, 12 192x256, 32 64x64 small-M,
    256 128x128 8-wave deep-ring one-round kernel, 224 / 192 the fill-the-round 224x128 / 192x128 ping-pong kernel, 16 the 256x256 ping-pong
    kernel on v_mfma_f32_16x16x32_bf16 (the one variant with other last bits) (include/vl2hip.h)."""
    _CTL["variant"] = int(v)


def gemm(a, w, bias=None, res=None, act=ACT_NONE, swiglu=False, out_f32=False, out=None, M=None,
         gather=None, out_map=None, res_map=None, flop_k=None, stats_out=None, norm=None, norm_out=None, mfma16=False, tile_ctr=None):
    """C = epilogue(a @ w.T).  a [M,K] bf16 (or row pool when `gather`), w [N,K] bf16, bias fp32 [N], res bf16 rows.
    gather = (a_idx int32 [nseg, M], zero_row (unused), seg_k).  out_map = (grp, grp_pad, row_off),
    res_map = (row_mod, row_off) -- see include/vl2hip.h.
    stats_out: fp32 [M, N/64, 2] buffer the epilogue fills with (sum, sum of squares) per row and 64-column block of the
    stored output.  norm = (kind, stats_in, eps, w_colsum): Norm(a) @ w.T computed on the raw rows of `a` from the statistics
    the GEMM that wrote `a` emitted (NORM_RMS / NORM_LN; w must carry the norm weight folded in, bias the folded shift).
    norm_out = (kind, eps, row_norm_out fp32 [M, 2], tickets uint32 [>= M/64 + 2], zeroed once): with stats_out, the call also leaves
    (mean, rstd) of its OUTPUT rows in row_norm_out -- what row_norm_finalize(stats_out) would compute, without the launch."""
    _chk(a, _lib.elem_dtype(), "a"); _chk(w, _lib.elem_dtype(), "w"); _chk(bias, torch.float32, "bias"); _chk(res, _lib.elem_dtype(), "res")
    N = w.shape[0]
    if gather is not None:
        a_idx, _zero_row, seg_k = gather
        K = seg_k * a_idx.shape[0]
        M = a_idx.shape[1] if M is None else M
    else:
        a_idx = None
        seg_k = 0
        K = w.shape[1]
        M = a.shape[0] if M is None else M
        if a.shape[1] != K:
            raise ValueError(f"gemm: a is [{a.shape[0]},{a.shape[1]}] but w is [{N},{K}]")
    ncol = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, ncol), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=a.device)
    grp, grp_pad, row_off = out_map or (0, 0, 0)
    rmod, roff = res_map or (0, 0)
    ws = _ws(a.device)
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0) | (GEMM_SPLITK if (_CTL["splitk"] and ws is not None) else 0) | _CTL["gemm_flags"] | \
            (GEMM_MFMA16 if ((mfma16 and not _CTL["stage_flags"] & STAGE_NO_MFMA16) or (swiglu and _CTL["stage_flags"] & STAGE_MFMA16)) else 0)
    kind, stats_in, eps, colsum = norm if norm is not None else (NORM_NONE, None, 0.0, None)
    row_norm = None
    if stats_in is not None and stats_in.dim() == 2:      # [rows, 2] = already reduced (row_norm_finalize)
        row_norm, stats_in = stats_in, None
        if row_norm.shape[0] < M or row_norm.shape[1] != 2:
            raise ValueError(f"gemm: row_norm must be [>={M}, 2], got {tuple(row_norm.shape)}")
        _chk(row_norm, torch.float32, "row_norm")
    if stats_out is not None and tuple(stats_out.shape) != (M, N // 64, 2):
        raise ValueError(f"gemm: stats_out must be [{M}, {N // 64}, 2], got {tuple(stats_out.shape)}")
    if stats_in is not None and (stats_in.shape[0] < M or tuple(stats_in.shape[1:]) != (K // 64, 2)):
        raise ValueError(f"gemm: stats_in must be [>={M}, {K // 64}, 2], got {tuple(stats_in.shape)}")
    _chk(stats_out, torch.float32, "stats_out"); _chk(stats_in, torch.float32, "stats_in"); _chk(colsum, torch.float32, "w_colsum")
    d = _lib.GemmDesc(ctypes.sizeof(_lib.GemmDesc), M, N, K, _p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0),
                      _p(bias), _p(res), res.stride(0) if res is not None else 0, act, flags, _p(a_idx), seg_k,
                      grp, grp_pad, row_off, rmod, roff, _p(stats_out), _p(stats_in), kind, float(eps), _p(colsum), _p(row_norm),
                      _p(ws), ws.numel() if ws is not None else 0, _CTL["variant"])
    if norm_out is not None:
        ko, eo, rn_out, tick = norm_out
        _chk(rn_out, torch.float32, "row_norm_out")
        if stats_out is None or rn_out.shape[0] < M or tick.numel() < M // 64 + 2 or tick.dtype not in (torch.int32, torch.uint32):
            raise ValueError("gemm: norm_out needs stats_out, row_norm_out [>= M, 2] and >= M/64 + 2 zeroed 32-bit tickets")
        d.row_norm_out, d.row_ticket, d.norm_out, d.norm_out_eps = _p(rn_out), _p(tick), ko, float(eo)
    if tile_ctr is not None:        # >= 16 bytes, zeroed once: the tile queue of the persistent form (GEMM_PERSISTENT; the kernels re-arm it) -- what the stage calls pass
        d.tile_ctr = _p(tile_ctr)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vl2_gemm", ctypes.byref(d), _stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("gemm", 2.0 * M * N * (flop_k or K), e0, e1, (M, N, K)))
    return out


def row_norm_finalize(stats, K, kind, eps, out=None):
    """stats [rows, K/64, 2] -> [rows, 2] = (mean, rstd) for NORM_RMS / NORM_LN: hand the result to gemm(norm=(kind, <it>, eps,
    colsum)) so the consuming GEMM does not reduce the partials again in every column tile."""
    _chk(stats, torch.float32, "stats")
    rows, np_ = stats.shape[0], stats.shape[1]
    out = torch.empty((rows, 2), dtype=torch.float32, device=stats.device) if out is None else out
    _lib.call("vl2_row_norm_finalize", _p(stats), _p(out), rows, np_, K, kind, float(eps), _stream())
    return out


def row_stats(x, out=None):
    """(sum, sum of squares) per row and 64-column block of x [rows, C] bf16 -> fp32 [rows, C/64, 2]: seeds a norm-carrying
    GEMM chain for a tensor no GEMM wrote (same layout and summation order as `gemm(..., stats_out=)`)."""
    _chk(x, _lib.elem_dtype(), "x")
    rows, C = x.shape
    out = torch.empty((rows, C // 64, 2), dtype=torch.float32, device=x.device) if out is None else out
    _lib.call("vl2_row_stats", _p(x), _p(out), rows, C, x.stride(0), _stream())
    return out


def layernorm(x, w, b, eps, res=None, silu=False, out=None):
    _chk(x, _lib.elem_dtype(), "x"); _chk(w, torch.float32, "w"); _chk(b, torch.float32, "b"); _chk(res, _lib.elem_dtype(), "res")
    rows, C = x.shape
    out = torch.empty((rows, C), dtype=_lib.elem_dtype(), device=x.device) if out is None else out
    _lib.call("vl2_layernorm", _p(x), _p(out), _p(w), _p(b), _p(res), rows, C, x.stride(0), out.stride(0),
              res.stride(0) if res is not None else 0, float(eps), int(silu), _stream())
    return out


def rmsnorm(x, w, eps, out=None):
    _chk(x, _lib.elem_dtype(), "x"); _chk(w, torch.float32, "w")
    rows, C = x.shape
    out = torch.empty((rows, C), dtype=_lib.elem_dtype(), device=x.device) if out is None else out
    _lib.call("vl2_rmsnorm", _p(x), _p(out), _p(w), rows, C, x.stride(0), out.stride(0), float(eps), _stream())
    return out


_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}       # frame types of the patch-row kernels (any build reads all three)


def patchify(frames, patch, kp):
    """frames [T,3,H,W] fp32/fp16/bf16 -> [T*G*G, kp] bf16 im2col rows."""
    if frames.dtype not in _DTYPE_CODE:
        raise TypeError(f"frames dtype {frames.dtype} not supported")
    frames = frames.contiguous()
    T, C, H, W = frames.shape
    G = H // patch
    out = torch.empty((T * G * G, kp), dtype=_lib.elem_dtype(), device=frames.device)
    _lib.call("vl2_patchify", _p(frames), _DTYPE_CODE[frames.dtype], _p(out), T, H, W, patch, G, kp, _stream())
    return out


def patchify_u8(frames_thwc, patch, kp, rescale, mean, std):
    """frames [T,H,W,3] uint8 (device) -> [T*G*G, kp] bf16 im2col rows of (x*rescale - mean) / std."""
    _chk(frames_thwc, torch.uint8, "frames")
    frames_thwc = frames_thwc.contiguous()
    T, H, W, C = frames_thwc.shape
    if C != 3:
        raise ValueError(f"expected uint8 frames [T,H,W,3], got {tuple(frames_thwc.shape)}")
    G = H // patch
    out = torch.empty((T * G * G, kp), dtype=_lib.elem_dtype(), device=frames_thwc.device)
    m, sd = [float(v) for v in mean], [float(v) for v in std]
    _lib.call("vl2_patchify_u8", _p(frames_thwc), _p(out), T, H, W, patch, G, kp, float(rescale), m[0], m[1], m[2], sd[0], sd[1], sd[2],
              _stream())
    return out


def fill_cls(x, cls_pos, T, rows_per_frame):
    _lib.call("vl2_fill_cls", _p(x), _p(cls_pos), T, x.shape[1], rows_per_frame, _stream())


def attn_fwd(q, k, v, o, q_str, k_str, v_str, o_str, B, H, nq, nk, group, scale, causal, causal_off, D):
    """Strided fused attention; *_str = (batch_stride, head_stride, row_stride) in elements; q/k/v/o may be views
    into one fused buffer (pass tensors whose data_ptr is the first element of head 0, batch 0)."""
    _lib.call("vl2_attn_fwd", _p(q), _p(k), _p(v), _p(o), *q_str, *k_str, *v_str, *o_str, B, H, nq, nk, group, float(scale),
              int(causal), causal_off, D, _CTL["attn_variant"], _stream())
    return o


def dwconv3x3_ln_silu(x, w9c, lnw, lnb, F, H, W, eps=1e-5):
    _chk(x, _lib.elem_dtype(), "x")
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.call("vl2_dwconv3x3_ln_silu", _p(x), _p(y), _p(w9c), _p(lnw), _p(lnb), F, H, W, C, float(eps), _stream())
    return y


def dwconv3x3_ln_silu_mean(x, w9c, lnw, lnb, F, H, W, eps=1e-5):
    """Depthwise 3x3 + LayerNorm2d + SiLU AND the SE squeeze of the result: returns (y [F*H*W, C], mean fp32 [F, C])."""
    _chk(x, _lib.elem_dtype(), "x")
    C = x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty((F, C), dtype=torch.float32, device=x.device)
    nb = _lib.load().vl2_dwconv_mean_workspace_bytes(F, C)
    ws = torch.empty((nb,), dtype=torch.uint8, device=x.device)
    _lib.call("vl2_dwconv3x3_ln_silu_mean", _p(x), _p(y), _p(w9c), _p(lnw), _p(lnb), F, H, W, C, float(eps), _p(mean), _p(ws), nb, _stream())
    return y, mean


def se_excite_scale_(x, g1, w2, b2, F, HW):
    """x[f, :, c] *= sigmoid(w2[c] . g1[f] + b2[c]) in place (SE excite + scale, one launch)."""
    _chk(x, _lib.elem_dtype(), "x"); _chk(g1, torch.float32, "g1"); _chk(w2, _lib.elem_dtype(), "w2"); _chk(b2, torch.float32, "b2")
    _lib.call("vl2_se_excite_scale", _p(x), _p(g1), _p(w2), _p(b2), F, HW, x.shape[-1], w2.shape[1], _stream())
    return x


def chan_mean(x, F, HW):
    C = x.shape[-1]
    m = torch.empty((F, C), dtype=torch.float32, device=x.device)
    _lib.call("vl2_chan_mean", _p(x), _p(m), F, HW, C, _stream())
    return m


def small_linear(x, w, b, act):
    _chk(x, torch.float32, "x"); _chk(w, _lib.elem_dtype(), "w")
    F, K = x.shape
    N = w.shape[0]
    out = torch.empty((F, N), dtype=torch.float32, device=x.device)
    _lib.call("vl2_small_linear", _p(x), _p(w), _p(b), _p(out), F, N, K, act, _stream())
    return out


def se_scale_(x, gate, F, HW):
    _lib.call("vl2_se_scale", _p(x), _p(gate), F, HW, x.shape[-1], _stream())
    return x


def rope_kv(qkv, q_out, kcache, vcache, cos_t, sin_t, nh, nkv, pos0):
    S = qkv.shape[0]
    smax = kcache.shape[1]
    _lib.call("vl2_rope_kv", _p(qkv), _p(q_out), _p(kcache), _p(vcache), _p(cos_t), _p(sin_t), S, nh, nkv, smax, pos0, _stream())


def gemv(w, x, norm_w=None, eps=1e-5, res=None, swiglu=False, out_f32=False, out=None, bias=None):
    _chk(w, _lib.elem_dtype(), "w"); _chk(x, _lib.elem_dtype(), "x"); _chk(bias, torch.float32, "bias")
    N, K = w.shape
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((n_out,), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=w.device)
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0)
    _lib.call("vl2_gemv_bf16", _p(w), _p(x), _p(norm_w), _p(res), _p(bias), _p(out), N, K, w.stride(0), float(eps), flags, _stream())
    return out


def quant_fp8(w):
    """include/vl2hip.h vl2_pack_quant_fp8: 16-bit weights [N, K] -> (e4m3fn bytes [N, K] as uint8, power-of-two row scales [N] fp32)."""
    _chk(w, _lib.elem_dtype(), "w")
    N, K = w.shape
    q = torch.empty((N, K), dtype=torch.uint8, device=w.device)
    sc = torch.empty((N,), dtype=torch.float32, device=w.device)
    _lib.call("vl2_pack_quant_fp8", _p(w), N, K, w.stride(0), _p(q), _p(sc), _stream())
    return q, sc


def quant_act_fp8(x, rms_eps=None, out=None, row_tab=None):
    """include/vl2hip.h vl2_quant_act_fp8: 16-bit activation rows [M, K] -> (e4m3fn bytes [M, K] as uint8, row table [M, 2] fp32 = (0, row scale
    [x RMS rstd when rms_eps is given])), the A operand and `row_norm` of gemm_fp8."""
    _chk(x, _lib.elem_dtype(), "x")
    M, K = x.shape
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device) if out is None else out
    tab = torch.empty((M, 2), dtype=torch.float32, device=x.device) if row_tab is None else row_tab
    _lib.call("vl2_quant_act_fp8", _p(x), x.stride(0), _p(q), q.stride(0), _p(tab), M, K, NORM_NONE if rms_eps is None else NORM_RMS,
              0.0 if rms_eps is None else float(rms_eps), _stream())
    return q, tab


def gemm_fp8(a8, row_tab, w8, col_scale, bias=None, res=None, act=ACT_NONE, swiglu=False, out_f32=False, out=None):
    """C = epilogue(row_tab[m, 1] * col_scale[n] * (a8 @ w8.T)) on the fp8 matrix pipe (vl2_gemm with VL2_GEMM_FP8; W8A8, an OPTIONAL
    arithmetic): a8 [M, K] / w8 [N, K] e4m3fn bytes (quant_act_fp8 / quant_fp8), K % 128 == 0, N % 256 == 0."""
    _chk(a8, torch.uint8, "a8"); _chk(w8, torch.uint8, "w8"); _chk(row_tab, torch.float32, "row_tab"); _chk(col_scale, torch.float32, "col_scale")
    _chk(bias, torch.float32, "bias"); _chk(res, _lib.elem_dtype(), "res")
    M, K = a8.shape
    N = w8.shape[0]
    if w8.shape[1] != K or row_tab.shape[0] < M or col_scale.shape[0] != N:
        raise ValueError("gemm_fp8: shapes of a8 / w8 / row_tab / col_scale do not agree")
    ncol = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, ncol), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=a8.device)
    flags = GEMM_FP8 | (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0)
    d = _lib.GemmDesc(ctypes.sizeof(_lib.GemmDesc), M, N, K, _p(a8), a8.stride(0), _p(w8), w8.stride(0), _p(out), out.stride(0),
                      _p(bias), _p(res), res.stride(0) if res is not None else 0, act, flags, None, 0, 0, 0, 0, 0, 0, None, None, NORM_NONE, 0.0, None,
                      _p(row_tab), None, 0, _CTL["variant"] if _CTL["variant"] in (4, 8, 12) else 0, _p(col_scale))
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("vl2_gemm", ctypes.byref(d), _stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("gemm_fp8", 2.0 * M * N * K, e0, e1, (M, N, K)))
    return out


def gemv_fp8(q, scale, x, norm_w=None, eps=1e-5, res=None, swiglu=False, out_f32=False, out=None, bias=None, rms_plain=False):
    """y = scale * (q @ x) (+ bias) (+ res) for one token on fp8 weights (include/vl2hip.h vl2_gemv_fp8; W8A16)."""
    _chk(q, torch.uint8, "q"); _chk(scale, torch.float32, "scale"); _chk(x, _lib.elem_dtype(), "x"); _chk(bias, torch.float32, "bias")
    N, K = q.shape
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((n_out,), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=q.device)
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0) | (GEMV_RMS_PLAIN if rms_plain else 0)
    _lib.call("vl2_gemv_fp8", _p(q), _p(scale), _p(x), _p(norm_w), _p(res), _p(bias), _p(out), N, K, q.stride(0), float(eps), flags, _stream())
    return out


def gemm_skinny(a, w, bias=None, res=None, swiglu=False, out_f32=False, out=None):
    """C = epilogue(a @ w.T) for M = a.shape[0] <= 64 rows (batched decode): weights streamed once, GEMV-style, into MFMA.
    Needs `attach_workspace` (fp32 partial sums of the K split)."""
    _chk(a, _lib.elem_dtype(), "a"); _chk(w, _lib.elem_dtype(), "w"); _chk(bias, torch.float32, "bias"); _chk(res, _lib.elem_dtype(), "res")
    M, K = a.shape
    N = w.shape[0]
    ncol = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, ncol), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=a.device)
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0)
    ws = attach_workspace(a.device)
    _lib.call("vl2_gemm_skinny_bf16", _p(a), _p(w), _p(out), _p(bias), _p(res), M, N, K, a.stride(0), w.stride(0), out.stride(0),
              0 if res is None else res.stride(0), flags, _p(ws), ws.numel(), _stream())
    return out


def gemv_batched(w, x, norm_w=None, eps=1e-5, res=None, swiglu=False, out_f32=False, out=None, bias=None):
    """y[b] = W x[b] for the rows of x [MB, K] in one pass over W (batched decode).  res / out are [MB, n_out]."""
    _chk(w, _lib.elem_dtype(), "w"); _chk(x, _lib.elem_dtype(), "x"); _chk(bias, torch.float32, "bias")
    N, K = w.shape
    MB = x.shape[0]
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((MB, n_out), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=w.device)
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_OUT_F32 if out_f32 else 0)
    _lib.call("vl2_gemv_batched_bf16", _p(w), _p(x), _p(norm_w), _p(res), _p(bias), _p(out), MB, N, K, w.stride(0), x.stride(0),
              out.stride(0), 0 if res is None else res.stride(0), float(eps), flags, _stream())
    return out


def attn_decode(qkv, kcache, vcache, cos_t, sin_t, partial, out, nh, nkv, pos, scale, pos_dev=None, ctx_cap=0):
    """RoPE + KV append + flash-decoding attention of one new token (un-roped fused qkv row) at position `pos`
    (or *pos_dev, for hipGraph replay; then ctx_cap bounds the positions the launch covers)."""
    _lib.call("vl2_attn_decode", _p(qkv), _p(kcache), _p(vcache), _p(cos_t), _p(sin_t), _p(partial), _p(out), nh, nkv,
              kcache.shape[1], int(pos), _p(pos_dev), int(ctx_cap), float(scale), _stream())
    return out


def attn_decode_fused(qkv, kcache, vcache, cos_t, sin_t, partial, out, nh, nkv, pos_dev, scale, cnt):
    """attention + combine of one decode token in ONE launch (vl2_attn_decode_fused): position from device memory, `partial` sized for
    the whole cache, `cnt` = nkv zeroed int32 ticket counters (re-zero before every call).  Same bits as `attn_decode`."""
    smax = kcache.shape[-2]
    _lib.call("vl2_attn_decode_fused", _p(qkv), _p(kcache), _p(vcache), _p(cos_t), _p(sin_t), _p(partial), _p(out), nh, nkv, smax,
              _p(pos_dev), float(scale), _p(cnt), _stream())
    return out


def attn_decode_batched(qkv, kcache, vcache, cos_t, sin_t, partial, out, nh, nkv, pos_dev, ctx_cap, scale):
    """Batched decode attention: qkv [B, (nh+2nkv)*128], caches [B, nkv, smax, 128], out [B, nh*128], pos_dev int32 [B]."""
    B = qkv.shape[0]
    _lib.call("vl2_attn_decode_batched", _p(qkv), _p(kcache), _p(vcache), _p(cos_t), _p(sin_t), _p(partial), _p(out), B,
              qkv.stride(0), kcache.stride(0), out.stride(0), nh, nkv, kcache.shape[2], _p(pos_dev), int(ctx_cap), float(scale), _stream())
    return out


def argmax(logits, tok, hist=None, step=0, state=None):
    _lib.call("vl2_argmax", _p(logits), logits.numel(), _p(tok), _p(hist), step, _p(state), _stream())


def sample_token(logits, tok, u, temperature, top_k=50, top_p=1.0, hist=None, step=0, state=None, dbg=None):
    """One sampled token from fp32 logits (include/vl2hip.h vl2_sample_token): temperature -> top-k -> top-p (HF's warpers, HF's order), then the
    inverse CDF of the kept, renormalised probabilities at u[step] (u: device fp32 uniforms in [0, 1)).  tok / hist / step / state as `argmax`."""
    _chk(logits, torch.float32, "logits"); _chk(u, torch.float32, "u"); _chk(dbg, torch.float32, "dbg")
    _lib.call("vl2_sample_token", _p(logits), logits.numel(), float(temperature), int(top_k), float(top_p), _p(u), _p(tok), _p(hist), int(step),
              _p(state), _p(dbg), _stream())


def embed_rows(ids_i32, table, out):
    _lib.call("vl2_embed_rows", _p(ids_i32), _p(table), _p(out), ids_i32.numel(), table.shape[1], out.stride(0), _stream())
    return out


# ------------------------------------------------------------------------------------------------ one-time weight re-layout
# The C-ABI forms of what weights.py does with tensor ops (include/vl2hip.h vl2_pack_*): for hosts without PyTorch; here they are
# bound so that the tests can hold them against weights.py byte for byte.
def pack_fold_norm(w, g, beta=None, bias=None):
    """(W' bf16 [N,K], colsum fp32 [N], shift fp32 [N] or None) = weights.fold_norm(w, g, beta, bias)."""
    for t, n in ((w, "w"), (g, "g"), (beta, "beta"), (bias, "bias")):
        _chk(t, _lib.elem_dtype(), n)
    N, K = w.shape
    wp = torch.empty((N, K), dtype=_lib.elem_dtype(), device=w.device)
    s = torch.empty((N,), dtype=torch.float32, device=w.device)
    t = torch.empty((N,), dtype=torch.float32, device=w.device) if beta is not None else None
    _lib.call("vl2_pack_fold_norm", _p(w), _p(g), _p(beta), _p(bias), _p(wp), _p(s), _p(t), N, K, w.stride(0), _stream())
    return wp, s, t


def pack_gate_up(gate, up):
    _chk(gate, _lib.elem_dtype(), "gate"); _chk(up, _lib.elem_dtype(), "up")
    I, D = gate.shape
    out = torch.empty((2 * I, D), dtype=_lib.elem_dtype(), device=gate.device)
    _lib.call("vl2_pack_gate_up", _p(gate), _p(up), _p(out), I, D, _stream())
    return out


def pack_permute(x, out_f32=False):
    """[A, B, C] bf16 -> [A, C, B] (bf16, or fp32 with out_f32)."""
    _chk(x, _lib.elem_dtype(), "x")
    A, B, C = x.shape
    out = torch.empty((A, C, B), dtype=torch.float32 if out_f32 else _lib.elem_dtype(), device=x.device)
    _lib.call("vl2_pack_permute", _p(x), _p(out), A, B, C, int(out_f32), _stream())
    return out


def pack_pad_rows(x, cols_dst):
    _chk(x, _lib.elem_dtype(), "x")
    rows, cs = x.shape
    out = torch.empty((rows, cols_dst), dtype=_lib.elem_dtype(), device=x.device)
    _lib.call("vl2_pack_pad_rows", _p(x), _p(out), rows, cs, cols_dst, _stream())
    return out


def pack_cvt_f32(x):
    _chk(x, _lib.elem_dtype(), "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.call("vl2_pack_cvt_f32", _p(x), _p(out), x.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------------ stage-level entry points
# One C call per stage (include/vl2hip.h vl2_vit_forward / vl2_llm_prefill / vl2_llm_decode_step): the layer loop runs inside
# libvl2hip.so.  The descriptors hold raw device pointers, so every builder returns (desc, keepalive): the caller keeps
# `keepalive` (the tensors and the ctypes layer array) referenced for as long as the descriptor is used.
STAGE_ABI = True        # the towers / connector / decoder use the stage calls ...


def stage_enabled():
    """... unless a per-GEMM profile is being taken (ops.PROFILE) or one of the per-call experiment controls (forced GEMM
    variant, split-K, forced attention variant: set_gemm_variant / set_splitk / set_attn_kv_groups) is on: those are arguments of
    the per-operator entry points, the stage calls always use the library's own choice."""
    return STAGE_ABI and PROFILE is None and not (_CTL["variant"] or _CTL["splitk"] or _CTL["attn_variant"])


def vit_desc(w, v, family, act):
    """weights.pack_tower / pack_siglip_tower dict + vision config -> (_lib.VitDesc, keepalive)."""
    layers = (_lib.VitLayer * len(w["layers"]))()
    for i, lw in enumerate(w["layers"]):
        for n in ("wqkv", "bqkv", "sqkv", "wo", "bo", "w1", "b1", "s1", "w2", "b2"):
            setattr(layers[i], n, _p(lw[n]))
    nh = v["num_attention_heads"]
    hd_real = v["hidden_size"] // nh
    d = _lib.VitDesc(ctypes.sizeof(_lib.VitDesc), family, v["image_size"], v["patch_size"], v["hidden_size"], w["layers"][0]["w1"].shape[0], nh,
                     w.get("hdp", hd_real), len(w["layers"]), w["kp"], act, float(v["layer_norm_eps"]), float(hd_real ** -0.5),
                     _p(w["patch_w"]), _p(w.get("patch_b")), _p(w["pos"]), _p(w.get("cls_pos")), _p(w.get("pre_w")), _p(w.get("pre_b")), layers)
    return d, (layers, w)


def vit_forward(desc, frames, T, out, u8_norm=None):
    """frames [T,3,S,S] fp32 / fp16 / bf16 or uint8 [T,S,S,3] -> out [T * tokens, D] bf16 (class-token row included)."""
    code = 3 if frames.dtype == torch.uint8 else _DTYPE_CODE[frames.dtype]
    n = int(_lib.load().vl2_vit_workspace_bytes(ctypes.byref(desc), T))
    if n < 0:
        raise _lib.Vl2HipError("vl2_vit_workspace_bytes: bad descriptor")
    ws = torch.empty((n,), dtype=torch.uint8, device=out.device)
    nrm = (ctypes.c_float * 7)(*[float(x) for x in u8_norm]) if u8_norm is not None else None
    desc.flags = _CTL["stage_flags"]
    _lib.call("vl2_vit_forward", ctypes.byref(desc), _p(frames), code, ctypes.addressof(nrm) if nrm is not None else None, T, _p(out),
              _p(ws), n, _stream())
    return out


def stc_desc(w):
    """weights.pack_connector dict -> (_lib.StcDesc, keepalive)."""
    d = _lib.StcDesc()
    d.size = ctypes.sizeof(_lib.StcDesc)
    d.cin, d.C = w["s1"][0]["conv1_w"].shape[1], w["s1"][0]["conv1_w"].shape[0]
    for stage, blocks in (("s1", d.s1), ("s2", d.s2)):
        for i, b in enumerate(w[stage]):
            for n, _t in _lib.StcBlock._fields_[:-1]:
                setattr(blocks[i], n, _p(b.get(n)))
            blocks[i].rd = b["fc1_w"].shape[0]
    for n in ("samp_w", "samp_b", "ro0_w", "ro0_b", "ro2_w", "ro2_b"):
        setattr(d, n, _p(w[n]))
    return d, (w,)


def stc_forward(desc, x, T, hw, idx, dims, out):
    """x [T*hw*hw, cin] bf16 tower features -> out [To*Ho*Wo, C] bf16 (idx, dims from connector.conv3d_k2s2p1_index)."""
    _chk(x, _lib.elem_dtype(), "x"); _chk(out, _lib.elem_dtype(), "out"); _chk(idx, torch.int32, "idx")
    To, Ho, Wo = dims
    n = int(_lib.load().vl2_stc_workspace_bytes(ctypes.byref(desc), T, hw, To * Ho * Wo))
    if n < 0:
        raise _lib.Vl2HipError("vl2_stc_workspace_bytes: bad descriptor")
    ws = torch.empty((n,), dtype=torch.uint8, device=out.device)
    desc.flags = _CTL["stage_flags"]
    _lib.call("vl2_stc_forward", ctypes.byref(desc), _p(x), T, hw, _p(idx), To, Ho, Wo, _p(out), _p(ws), n, _stream())
    return out


def llm_desc(w, cfg_llm, nh, nkv, smax, eps, kcache, vcache, cos_t, sin_t, w8=None):
    """weights.pack_decoder dict + this decoder's caches -> (_lib.LlmDesc, keepalive).  w8: optional fp8 copies (decoder.enable_fp8_decode)."""
    layers = (_lib.LlmLayer * len(w["layers"]))()
    for i, lw in enumerate(w["layers"]):
        layers[i].wqkv, layers[i].bqkv, layers[i].wo = _p(lw["wqkv"]), _p(lw["bqkv"]), _p(lw["wo"])
        layers[i].wgu, layers[i].wd = _p(lw["wgu"]), _p(lw["wd"])
        layers[i].kcache, layers[i].vcache = _p(kcache[i]), _p(vcache[i])
    d = _lib.LlmDesc(ctypes.sizeof(_lib.LlmDesc), cfg_llm["hidden_size"], w["layers"][0]["wd"].shape[1], nh, nkv, len(w["layers"]),
                     w["lm_head"].shape[0], smax, float(eps), layers, _p(w["embed"]), _p(w["norm_w"]), _p(w["ones"]), _p(w["lm_head"]),
                     _p(cos_t), _p(sin_t))
    l8 = None
    if w8 is not None:
        l8 = (_lib.LlmLayerW8 * len(w["layers"]))()
        for i, q in enumerate(w8["layers"]):
            for n in ("qkv", "o", "gu", "d"):
                setattr(l8[i], "w" + n, _p(q["w" + n][0]))
                setattr(l8[i], "s" + n, _p(q["w" + n][1]))
        d.layers_w8 = l8
        d.lm_head_w8, d.lm_head_scale = _p(w8["lm_head"][0]), _p(w8["lm_head"][1])
    return d, (layers, w, kcache, vcache, l8, w8)


def _llm_ws(desc, S, device):
    n = int(_lib.load().vl2_llm_workspace_bytes(ctypes.byref(desc), S))
    if n < 0:
        raise _lib.Vl2HipError("vl2_llm_workspace_bytes: bad descriptor")
    return torch.empty((n,), dtype=torch.uint8, device=device), n


def llm_prefill(desc, x, logits_out, fp8=False):
    desc.flags = _CTL["stage_flags"] | (STAGE_PREFILL_FP8 if fp8 else 0)      # (before the size query: the fp8 activation image is carved only with the flag)
    ws, n = _llm_ws(desc, x.shape[0], x.device)
    _lib.call("vl2_llm_prefill", ctypes.byref(desc), _p(x), x.shape[0], _p(logits_out), _p(ws), n, _stream())
    return logits_out


def llm_decode_step(desc, logits, tok, state, hist, partial, ws, fp8=False):
    desc.flags = _CTL["stage_flags"] | (STAGE_DECODE_FP8 if fp8 else 0)
    _lib.call("vl2_llm_decode_step", ctypes.byref(desc), _p(logits), _p(tok), _p(state), _p(hist), _p(partial), _p(ws), ws.numel(), _stream())
