"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the fp8 weight format of the optional decode path (the fp8 half of
BASELINE.json configs[4]; product: videollama2_amd/csrc/k_fp8.h, include/vl2hip.h vl2_pack_quant_fp8 / vl2_gemv_fp8).

PARITY UNPINNED against the reference: /root/reference has no fp8 path and no calibration recipe ("needs a calibration story
the reference does not have"), so this file DEFINES the arithmetic and the kernels are held bit for bit (quantiser) / to fp32 rounding (GEMV)
against it.  What it is pinned to instead: the OCP 8-bit floating point specification's e4m3fn encoding (OFP8 v1.0: bias 7, no infinities,
S.1111.111 = NaN, max finite 448, subnormals in units of 2^-9) through (a) the known-answer table KNOWN_E4M3FN below, (b) PyTorch's
independent implementation of the same format (`torch.float8_e4m3fn`), checked byte for byte over all 256 codes and over random inputs in
tests/test_fp8.py, (c) a third independent implementation, the emulator's (tests/emu/hip_emu.h).

Only tests/ and bench.py's checker legs may import this file (tests/test_host_logic.py::test_product_never_imports_oracle)."""
import math

import torch

# (value, code): OFP8 e4m3fn known answers -- exact values, the two ends of the normal range, subnormals, round-half-even ties, saturation
KNOWN_E4M3FN = [(0.0, 0x00), (1.0, 0x38), (-2.0, 0xC0), (0.5, 0x30), (448.0, 0x7E), (-448.0, 0xFE), (2.0 ** -6, 0x08), (2.0 ** -9, 0x01),
                (7 * 2.0 ** -9, 0x07), (1.125, 0x39), (1.0625, 0x38), (1.1875, 0x3A), (17.0, 0x58), (19.0, 0x5A), (240.0, 0x77),
                (3 * 2.0 ** -10, 0x02), (2.0 ** -10, 0x00), (0.75 * 2.0 ** -9, 0x01)]


def e4m3fn_decode(code):
    """One byte -> float (python arithmetic, from the specification)."""
    s, e, m = code >> 7, (code >> 3) & 15, code & 7
    if e == 15 and m == 7:
        return math.nan
    v = m * 2.0 ** -9 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7)
    return -v if s else v


def e4m3fn_encode(x):
    """float -> byte, round to nearest even, saturating to +-448 (python arithmetic, from the specification; exact for fp32 inputs)."""
    if x != x:
        return 0x7F
    s = 0x80 if math.copysign(1.0, x) < 0 else 0
    a = abs(x)
    best, bd = 0, None
    for c in range(0x7F):                      # the 127 finite non-negative codes are monotone: nearest value, ties to the even code
        d = abs(e4m3fn_decode(c) - a)
        if bd is None or d < bd or (d == bd and c % 2 == 0):
            best, bd = c, d
    return s | best


def row_scale_exponent(amax):
    """e = the smallest integer with amax <= 448 * 2^e (0 for amax = 0 or denormal amax), clamped to [-100, 100] -- k_fp8.h's bit arithmetic
    restated with frexp: amax = m * 2^x, m in [0.5, 1); 448 = 0.875 * 2^9."""
    amax = amax.float()
    m, x = torch.frexp(amax)
    e = x - 9 + (m > 0.875).to(x.dtype)
    e = torch.where(amax < 2.0 ** -126, torch.zeros_like(e), e)
    return e.clamp(-100, 100)


def quant_rows(w):
    """w [N, K] (any float dtype; the values are taken as they are) -> (q uint8 [N, K] e4m3fn codes, scale fp32 [N] = 2^e)."""
    wf = w.detach().float().cpu()
    e = row_scale_exponent(wf.abs().amax(dim=1))
    scale = torch.exp2(e.float())
    q = (wf * torch.exp2(-e.float())[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)     # the scaling is exact (a power of two)
    return q, scale


def dequant(q, scale):
    return q.view(torch.float8_e4m3fn).float() * scale[:, None]


def gemv(q, scale, x, norm_w=None, eps=1e-5, res=None, bias=None, swiglu=False, rms=False, elem=torch.bfloat16):
    """y = scale * (q x) (+ bias) (+ res) in fp32, x [K] in the 16-bit element type.  rms / norm_w: MistralRMSNorm on x first, its output
    rounded to the element type (HF:modeling_mistral.py:46-48).  swiglu: rows in blocks of 64 = 32 gate rows then 32 up rows."""
    xf = x.detach().float().cpu()
    if rms or norm_w is not None:
        xf = xf * torch.rsqrt((xf * xf).mean() + eps)
        if norm_w is not None:
            xf = xf * norm_w.float().cpu()
        xf = xf.to(elem).float()
    y = (q.view(torch.float8_e4m3fn).float() @ xf) * scale
    if swiglu:
        y = y.view(-1, 2, 32)
        g, u = y[:, 0].reshape(-1), y[:, 1].reshape(-1)
        y = torch.nn.functional.silu(g) * u
    if bias is not None:
        y = y + bias.float().cpu()
    if res is not None:
        y = y + res.float().cpu()
    return y


# ---- W8A8: the prefill projections on the fp8 matrix pipe (include/vl2hip.h VL2_GEMM_FP8, vl2_quant_act_fp8; csrc/k_fp8.h quant_act_fp8_kernel,
# k_gemm.h gemm3 / gemm4 FP8).  PARITY UNPINNED against the reference for the same reason as above (BASELINE.json configs[4] names "fp8 MFMA on
# CDNA4"; the reference holds no fp8 arithmetic to compare with): this restatement is the definition.
def quant_act_rows(x, rms_eps=None):
    """x [M, K] (16-bit activations) -> (q uint8 [M, K], row table fp32 [M, 2] = (0, sa[m]) or (0, sa[m] * rsqrt(mean_k x^2 + eps))): the weight
    quantiser's power-of-two row scale rule applied to activation rows; the RMS factor is MistralRMSNorm's rstd of the RAW row
    (HF:modeling_mistral.py:46-48; the norm weight is folded into the projection)."""
    q, sa = quant_rows(x)
    xf = x.detach().float().cpu()
    rs = sa if rms_eps is None else sa * torch.rsqrt((xf * xf).mean(dim=1) + rms_eps)
    return q, torch.stack([torch.zeros_like(rs), rs], dim=1)


def gemm_w8a8(qa, tab, qw, sw, bias=None, res=None, act=None, swiglu=False):
    """C = epilogue(tab[m, 1] * sw[n] * sum_k qa[m, k] qw[n, k]) in fp32 (the 16-bit rounding of the stored result is the caller's): products of
    e4m3fn values are exact in fp32, the sum is fp32.  swiglu: columns in blocks of 64 = 32 gate then 32 up (weights.pack_gate_up)."""
    y = (qa.view(torch.float8_e4m3fn).float() @ qw.view(torch.float8_e4m3fn).float().T) * tab[:, 1:2].float().cpu() * sw.float().cpu()[None, :]
    if swiglu:
        M = y.shape[0]
        y = y.view(M, -1, 2, 32)
        y = (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, -1)
    if bias is not None:
        y = y + bias.float().cpu()[None, :]
    if act == "silu":
        y = torch.nn.functional.silu(y)
    if res is not None:
        y = y + res.float().cpu()
    return y


# ---- the W8A8 prefill as ONE chain (videollama2_amd/decoder.py prefill under enable_fp8_prefill; csrc/vl2_stage.inc VL2_STAGE_PREFILL_FP8) ----------
def _quant_rows_anywhere(w):
    """`quant_rows` on the device the tensor lives on (host cores or torch-ROCm; the conversion to torch.float8_e4m3fn is the same OFP8 rounding on
    both -- tests/test_gpu_parity_full.py checks the two against each other before it trusts the device)."""
    wf = w.detach().float()
    e = row_scale_exponent(wf.abs().amax(dim=1))
    return (wf * torch.exp2(-e.float())[:, None]).to(torch.float8_e4m3fn), torch.exp2(e.float())


def _w8a8(x, qw, sw, store, rms_eps=None):
    """One projection: the activation rows as `store` holds them -> e4m3fn codes with a power-of-two row scale (quant_act_rows; the RMS factor is
    the rstd of the RAW row), exact products, fp32 sum, x row scale x weight row scale.  Returns fp32 [M, N]."""
    xs = x.to(store)
    qa, sa = _quant_rows_anywhere(xs)
    if rms_eps is not None:
        xf = xs.float()
        sa = sa * torch.rsqrt((xf * xf).mean(dim=1) + rms_eps)
    return (qa.float() @ qw.float().T) * sa[:, None] * sw[None, :]


def quantise_decoder_for_prefill(sd, cfg, elem=torch.bfloat16, device="cpu", n_layers=None):
    """The four projections of every layer as the fp8 prefill streams them: W' = the packed matrix in the 16-bit element type (q/k/v and gate/up with
    the RMSNorm gain folded in and rounded, videollama2_amd/weights.py fold_norm:39-40), then one power-of-two scale per output row and e4m3fn codes."""
    n_layers = cfg["llm"]["num_hidden_layers"] if n_layers is None else n_layers
    out = []
    r = lambda w: w.to(device=device, dtype=elem).float()
    for i in range(n_layers):
        p = f"model.layers.{i}."
        g1, g2 = r(sd[p + "input_layernorm.weight"]), r(sd[p + "post_attention_layernorm.weight"])
        fold = lambda names, g: (torch.cat([r(sd[p + n + ".weight"]) for n in names], 0) * g[None, :]).to(elem)
        out.append(dict(qkv=_quant_rows_anywhere(fold(("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), g1)),
                        o=_quant_rows_anywhere(r(sd[p + "self_attn.o_proj.weight"])),
                        gate=_quant_rows_anywhere(fold(("mlp.gate_proj",), g2)), up=_quant_rows_anywhere(fold(("mlp.up_proj",), g2)),
                        down=_quant_rows_anywhere(r(sd[p + "mlp.down_proj.weight"]))))
    return out


def mistral_prefill_w8a8(sd, cfg, x, q8, elem=torch.bfloat16, chain=torch.float32):
    """Last-position logits of the prefill with W8A8 projections, as the product defines it: the quantisers read the activation in the 16-bit element
    type `elem` at the four places the product stores it (the residual stream before q/k/v and before gate/up, the attention output, the SwiGLU
    output); RoPE, the causal softmax attention (HF:modeling_mistral.py:96-117), the residual adds, the final norm and lm_head are the 16-bit path's.
    chain = torch.float32: the TRUTH of that definition (every other value kept in fp32); chain = elem: the same definition with every stored tensor
    rounded to the element type -- the floor a 16-bit implementation of it sits on.  q8 = quantise_decoder_for_prefill(...) on x's device."""
    import torch.nn.functional as F
    from oracle import vl2_oracle as O
    l = cfg["llm"]
    nh, nkv, hd, eps = l["num_attention_heads"], l["num_key_value_heads"], l["head_dim"], l["rms_norm_eps"]
    S, dev = x.shape[0], x.device
    cos, sin = O.rope_cos_sin(cfg, torch.arange(S), chain)
    cos, sin = cos.to(dev), sin.to(dev)
    x = x.to(chain)
    for w in q8:
        qkv = _w8a8(x, *w["qkv"], store=elem, rms_eps=eps).to(chain)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=1)
        q, k, v = q.view(S, nh, hd).transpose(0, 1), k.view(S, nkv, hd).transpose(0, 1), v.view(S, nkv, hd).transpose(0, 1)
        q, k = q * cos + O.rotate_half(q) * sin, k * cos + O.rotate_half(k) * sin
        rep = nh // nkv
        kk = k[:, None].expand(nkv, rep, S, hd).reshape(nh, S, hd)
        vv = v[:, None].expand(nkv, rep, S, hd).reshape(nh, S, hd)
        a = torch.matmul(q, kk.transpose(1, 2)) * (hd ** -0.5)
        a = a.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1)[None], torch.finfo(a.dtype).min)
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(chain)
        o = torch.matmul(a, vv).transpose(0, 1).reshape(S, nh * hd)
        x = (x.float() + _w8a8(o, *w["o"], store=elem)).to(chain)
        g, u = _w8a8(x, *w["gate"], store=elem, rms_eps=eps), _w8a8(x, *w["up"], store=elem, rms_eps=eps)
        act = (F.silu(g) * u).to(chain)
        x = (x.float() + _w8a8(act, *w["down"], store=elem)).to(chain)
    r = lambda t: t.to(device=dev, dtype=elem).to(chain)
    h = O.rmsnorm(x[-1:], r(sd["model.norm.weight"]), eps)
    return F.linear(h, r(sd["lm_head.weight"]))[0].float()
