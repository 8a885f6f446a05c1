"""fp8 (OCP e4m3fn) decode weights on MI355X -- SURVEY.md 8f row 5, the fp8 half of BASELINE.json configs[4] (the audio half has no source
in the reference tree).  The kernels of csrc/k_fp8.h through the C ABI against oracle/fp8_oracle.py: the quantiser bit for bit (hardware
v_cvt_pk_fp8_f32 == the OFP8 specification == PyTorch's float8_e4m3fn), the GEMV to fp32 rounding, the decode step against the SAME decoder
running its 16-bit kernels on the dequantised weights (isolates the kernels from the format), graph == eager, and the format's own error
against the unquantised decoder (reported -- an OPTIONAL arithmetic, never the default)."""
import json
import os

import pytest
import torch

from oracle import fp8_oracle as F8
from oracle import vl2_oracle as O
from tests.util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from videollama2_amd import _lib, ops as o
    _lib.load()
    return o


def bf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).bfloat16()


def test_quantiser_on_device_matches_the_oracle_bit_for_bit(ops):
    """Every e4m3fn boundary case in one matrix: all 127 midpoints (round-half-even ties) and points either side, subnormals, a zero row,
    tiny / huge rows, maxima on and just past 448 * 2^k; then a 7B-sized projection."""
    vals = []
    for c in range(0x7E):
        a, b = F8.e4m3fn_decode(c), F8.e4m3fn_decode(c + 1)
        vals += [a, (a + b) / 2, a + (b - a) * 0.25, a + (b - a) * 0.75]
    row = torch.tensor(vals + [448.0] * 8, dtype=torch.float32)
    row = torch.cat([row, -row])
    K = (row.numel() + 15) // 16 * 16
    w = torch.zeros(8, K)
    w[0, :row.numel()] = row                                     # scale 1: the values ARE the scaled values (where bf16 holds them exactly)
    w[1, :row.numel()] = row * 2.0 ** -12
    w[2, :row.numel()] = row * 2.0 ** 20
    w[4] = torch.randn(K, generator=torch.Generator().manual_seed(1)) * 0.02
    w[5, 0] = 450.0
    w[6] = torch.randn(K, generator=torch.Generator().manual_seed(2)) * 1e-30
    w = w.bfloat16()
    q, sc = ops.quant_fp8(w.to(DEV))
    qo, so = F8.quant_rows(w)
    assert torch.equal(sc.cpu(), so), (sc.cpu(), so)
    assert torch.equal(q.cpu(), qo), f"{int((q.cpu() != qo).sum())} codes differ"
    big = bf(4096, 14336, scale=14336 ** -0.5, seed=3)
    q, sc = ops.quant_fp8(big.to(DEV))
    qo, so = F8.quant_rows(big)
    assert torch.equal(sc.cpu(), so) and torch.equal(q.cpu(), qo)


@pytest.mark.parametrize("name,N,K,kw", [("qkv", 6144, 4096, dict(rms=True, bias=True)), ("o", 4096, 4096, dict(res=True)),
                                         ("gate_up", 28672, 4096, dict(rms=True, swiglu=True)), ("down", 4096, 14336, dict(res=True)),
                                         ("lm_head", 32000, 4096, dict(norm_w=True, f32=True)), ("qwen2_down", 3584, 18944, dict(res=True))])
def test_gemv_fp8_at_decoder_shapes(ops, name, N, K, kw):
    w, x = bf(N, K, scale=K ** -0.5, seed=1), bf(K, seed=2)
    q, sc = ops.quant_fp8(w.to(DEV))
    n_out = N // 2 if kw.get("swiglu") else N
    bias = torch.randn(n_out) if kw.get("bias") else None
    res = bf(n_out, seed=4) if kw.get("res") else None
    nw = (torch.rand(K) + 0.5) if kw.get("norm_w") else None
    y = ops.gemv_fp8(q, sc, x.to(DEV), norm_w=None if nw is None else nw.to(DEV), eps=1e-5, res=None if res is None else res.to(DEV),
                     bias=None if bias is None else bias.to(DEV), swiglu=bool(kw.get("swiglu")), out_f32=bool(kw.get("f32")),
                     rms_plain=bool(kw.get("rms")))
    ref = F8.gemv(q.cpu(), sc.cpu(), x, norm_w=nw, eps=1e-5, res=res, bias=bias, swiglu=bool(kw.get("swiglu")), rms=bool(kw.get("rms")))
    e = rel(y.float().cpu(), ref)
    assert e < (2e-4 if kw.get("f32") else 3e-3), (name, e)          # fp32 out: summation order + the 16-bit rounding of the normalised x falling either way; 16-bit out: one rounding


def test_fp8_decode_full_width_graph_equals_eager_and_tracks_dequantised_weights(ops):
    """Mistral-7B widths, 2 layers: S = 300 prefill (16-bit weights), then decode steps on the fp8 copies.  (1) captured hipGraph == eager
    loop, bit for bit; (2) == the same decoder's 16-bit kernels on the DEQUANTISED weights to rounding; (3) the format's own error against
    the unquantised weights, recorded in profiles/r04_fp8_parity.json."""
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    x = (torch.randn(300, 4096, generator=torch.Generator().manual_seed(2)).bfloat16().float() * 0.5).to(DEV)
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    t16, l16 = dec.generate(x, max_new_tokens=6, return_logits=True)
    dec.enable_fp8_decode()
    te, le = dec.generate(x, max_new_tokens=6, return_logits=True)
    tg, lg = dec.generate(x, max_new_tokens=6, return_logits=True, use_graph=True)
    assert te.tolist() == tg.tolist() and torch.equal(le, lg)
    assert torch.equal(le[0], l16[0])                              # the prefill logits do not involve the fp8 copies
    # (2) dequantised weights through the 16-bit kernels
    dec.enable_fp8_decode(False)
    for lw, q8 in zip(dec.w["layers"], dec.w8["layers"]):
        for k in ("wqkv", "wo", "wgu", "wd"):
            lw[k].copy_(F8.dequant(q8[k][0].cpu(), q8[k][1].cpu()).to(lw[k].dtype))
    dec.w["lm_head"].copy_(F8.dequant(dec.w8["lm_head"][0].cpu(), dec.w8["lm_head"][1].cpu()).to(dec.w["lm_head"].dtype))
    dec._stage = None
    td, ld = dec.generate(x, max_new_tokens=6, return_logits=True)  # (the prefill now also runs on the dequantised weights: compare step 1 on)
    rows = []
    # step 1 of the fp8 run fed the token of the (unquantised) prefill; feed the same token here by comparing only while the tokens agree
    same = 0
    for s in range(1, le.shape[0]):
        if te[0, :s].tolist() != t16[0, :s].tolist():
            break
        same = s
        rows.append(dict(step=s, fp8_vs_unquantised_rel_l2=rel(le[s].cpu(), l16[s].cpu()), top1_equal=bool(te[0, s] == t16[0, s])))
    assert same >= 1
    e_fmt = rows[0]["fp8_vs_unquantised_rel_l2"]
    print(f"[fp8] full-width decode, fp8 weights vs unquantised: step-1 logits rel-L2 {e_fmt:.3e}; tokens fp8 {te[0].tolist()} unquantised {t16[0].tolist()}")
    assert e_fmt < 0.15
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_fp8_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(dict(config="Mistral-7B widths, 2 layers, S=300 prefill on 16-bit weights, decode on e4m3fn row-scaled copies (W8A16)",
                   graph_equals_eager=True, rows=rows, tokens_fp8=te[0].tolist(), tokens_unquantised=t16[0].tolist()), open(out, "w"), indent=1)


def test_fp8_decode_kernels_equal_16bit_kernels_on_dequantised_weights(ops):
    """One decode step from an identical state: the fp8 projections against the 16-bit projections of the SAME decoder whose weights were
    replaced by the (bf16-exact) dequantised copies -- same products, fp32 summation order aside."""
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 6, only=keep)
    x = (torch.randn(200, 4096, generator=torch.Generator().manual_seed(3)).bfloat16().float() * 0.5).to(DEV)
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    dec.enable_fp8_decode()
    for lw, q8 in zip(dec.w["layers"], dec.w8["layers"]):           # BOTH runs prefill on the dequantised weights: identical caches
        for k in ("wqkv", "wo", "wgu", "wd"):
            lw[k].copy_(F8.dequant(q8[k][0].cpu(), q8[k][1].cpu()).to(lw[k].dtype))
    dec.w["lm_head"].copy_(F8.dequant(dec.w8["lm_head"][0].cpu(), dec.w8["lm_head"][1].cpu()).to(dec.w["lm_head"].dtype))
    t8, l8 = dec.generate(x, max_new_tokens=4, return_logits=True)
    dec.enable_fp8_decode(False)
    t16, l16 = dec.generate(x, max_new_tokens=4, return_logits=True)
    assert torch.equal(l8[0], l16[0])
    n = 1
    while n < l8.shape[0] and t8[0, :n].tolist() == t16[0, :n].tolist():
        e = rel(l8[n].cpu(), l16[n].cpu())
        print(f"[fp8] decode step {n}: fp8 kernels vs 16-bit kernels on dequantised weights rel-L2 {e:.2e}")
        assert e < 6e-3, (n, e)                                     # a 16-bit intermediate may round either way; fp32 logits stay together
        n += 1
    assert n >= 2


def test_fp8_kernels_in_the_fp16_build():
    """The fp16 build of the library (libvl2hip_f16.so): the quantiser reads half weights (every code of the same matrix as the bf16 case that
    half holds exactly), the GEMV takes half activations -- both against the oracle with `elem=torch.float16`."""
    from videollama2_amd import _lib, ops
    _lib.set_elem("fp16")
    try:
        g = torch.Generator().manual_seed(11)
        w = (torch.randn(512, 4096, generator=g) * 0.02).half()
        w[3] = 0
        w[5, 0] = 448.0
        q, sc = ops.quant_fp8(w.to(DEV))
        qo, so = F8.quant_rows(w)
        assert torch.equal(sc.cpu(), so) and torch.equal(q.cpu(), qo)
        x = (torch.randn(4096, generator=g)).half()
        y = ops.gemv_fp8(q, sc, x.to(DEV), eps=1e-5, out_f32=True, rms_plain=True)
        assert rel(y.cpu(), F8.gemv(qo, so, x, eps=1e-5, rms=True, elem=torch.float16)) < 2e-4
        ysw = ops.gemv_fp8(q, sc, x.to(DEV), eps=1e-5, swiglu=True, rms_plain=True)
        assert ysw.dtype == torch.float16
        assert rel(ysw.float().cpu(), F8.gemv(qo, so, x, eps=1e-5, rms=True, swiglu=True, elem=torch.float16)) < 2e-3
    finally:
        _lib.set_elem("bf16")


# ---------------------------------------------------------------------------------------------------------------------------------------
# W8A8 on the fp8 matrix pipe (v_mfma_f32_32x32x64_f8f6f4): VL2_GEMM_FP8 / vl2_quant_act_fp8 / VL2_STAGE_PREFILL_FP8 -- the MFMA half of
# BASELINE.json configs[4] ("fp8 MFMA on CDNA4").  Definition: oracle/fp8_oracle.py quant_act_rows / gemm_w8a8.
def test_activation_quantiser_on_device_matches_the_oracle(ops):
    """Token rows through vl2_quant_act_fp8: the bytes bit for bit (the weight quantiser's rule on activation rows: outliers, a zero row, tiny
    and huge rows), the row table (0, scale [x RMS rstd]) to fp32 rounding of the sum of squares."""
    x = bf(333, 4096, seed=7)
    x[0] = 0
    x[1, 5] = 200.0
    x[2] *= 1e-20
    x[3] *= 1e20
    for K in (4096, 14336):
        xx = x if K == 4096 else bf(97, K, seed=8)
        for eps in (None, 1e-5):
            q, tab = ops.quant_act_fp8(xx.to(DEV), rms_eps=eps)
            qo, to = F8.quant_act_rows(xx, rms_eps=eps)
            assert torch.equal(q.cpu(), qo), f"{int((q.cpu() != qo).sum())} codes differ (K={K})"
            assert torch.equal(tab[:, 0].cpu(), to[:, 0]) and rel(tab[:, 1].cpu(), to[:, 1]) < 1e-6


@pytest.mark.parametrize("name,M,N,K,kw", [("qkv", 1621, 6144, 4096, dict(rms=True, bias=True)), ("o", 1621, 4096, 4096, dict(res=True)),
                                           ("gate_up", 1621, 28672, 4096, dict(rms=True, swiglu=True)), ("down", 1621, 4096, 14336, dict(res=True)),
                                           ("down_ragged", 333, 4096, 14336, dict(res=True, variants=(4, 8, 12))),
                                           ("f32", 700, 1024, 2048, dict(f32=True, variants=(4, 8)))])
def test_gemm_fp8_at_decoder_shapes(ops, name, M, N, K, kw):
    """The four prefill projections at S = 1621 (and a ragged M on every tile shape) on the fp8 matrix pipe against gemm_w8a8: the same e4m3fn
    operands, exact products, fp32 sums in another order; a 16-bit output adds its one rounding.  The fp32-output bar is 5e-5, not the 1e-6 of
    a re-ordered IEEE sum: v_mfma_f32_32x32x64_f8f6f4 adds its 64 products per lane with a shared-exponent alignment that drops low bits
    (measured on MI355X at K = 2048: 1.45e-5 rel-L2 to the fp32 sum of the exact products, profiles/r05_experiments.md)."""
    x, w = bf(M, K, seed=1), bf(N, K, scale=K ** -0.5, seed=2)
    qw, sw = ops.quant_fp8(w.to(DEV))
    qa, tab = ops.quant_act_fp8(x.to(DEV), rms_eps=1e-5 if kw.get("rms") else None)
    n_out = N // 2 if kw.get("swiglu") else N
    bias = torch.randn(N) if kw.get("bias") else None
    res = bf(M, n_out, seed=4) if kw.get("res") else None
    ref = F8.gemm_w8a8(qa.cpu(), tab.cpu(), qw.cpu(), sw.cpu(), bias=bias, res=res, swiglu=bool(kw.get("swiglu")))
    try:
        for v in kw.get("variants", (0,)):
            ops.set_gemm_variant(v)
            y = ops.gemm_fp8(qa, tab, qw, sw, bias=None if bias is None else bias.to(DEV), res=None if res is None else res.to(DEV),
                             swiglu=bool(kw.get("swiglu")), out_f32=bool(kw.get("f32")))
            e = rel(y.float().cpu(), ref)
            assert e < (5e-5 if kw.get("f32") else 3e-3), (name, v, e)
    finally:
        ops.set_gemm_variant(0)


def test_fp8_prefill_full_width_stage_equals_operators_and_tracks_the_16bit_prefill(ops):
    """Mistral-7B widths, 2 layers, S = 300: the prefill with VL2_STAGE_PREFILL_FP8 (one C call) == decoder.prefill's operator loop bit for
    bit; against the 16-bit prefill the logits differ by the FORMAT (both operands rounded to e4m3fn): reported in
    profiles/r05_fp8_prefill_parity.json, bounded here, and switching the option off restores the 16-bit bits."""
    from videollama2_amd import ops as O_
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    x = (torch.randn(300, 4096, generator=torch.Generator().manual_seed(2)).bfloat16().float() * 0.5).to(DEV)
    dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
    l16 = dec.prefill(x).clone()
    dec.enable_fp8_prefill()
    try:
        outs = {}
        for stage in (True, False):
            O_.STAGE_ABI = stage
            outs[stage] = dec.prefill(x).clone()
    finally:
        O_.STAGE_ABI = True
    assert torch.equal(outs[True], outs[False])
    e = rel(outs[True].cpu(), l16.cpu())
    top1 = bool(int(outs[True].argmax()) == int(l16.argmax()))
    print(f"[fp8] full-width 2-layer prefill, W8A8 vs 16-bit: last-position logits rel-L2 {e:.3e}, top-1 equal {top1}")
    assert 1e-4 < e < 0.2
    dec.enable_fp8_prefill(False)
    assert torch.equal(dec.prefill(x), l16)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_fp8_prefill_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(dict(config="Mistral-7B widths, 2 layers, S=300: prefill projections W8A8 on v_mfma_f32_32x32x64_f8f6f4 vs the 16-bit prefill",
                   stage_equals_operator_loop=True, logits_rel_l2_vs_16bit=e, top1_equal=top1), open(out, "w"), indent=1)
