"""fp8 (OCP e4m3fn) decode weights, SURVEY.md 8f row 5 (the fp8 half of BASELINE.json configs[4]) -- CPU tests.
The reference has no fp8 path, so oracle/fp8_oracle.py DEFINES the arithmetic ("parity unpinned" against the reference, pinned to the OFP8
specification through three independent implementations: the oracle's python restatement + known answers, PyTorch's float8_e4m3fn, and the
emulator's).  The product kernels (csrc/k_fp8.h) are run through the emulator build here; tests/test_gpu_fp8.py runs them on MI355X."""
import math

import pytest
import torch

from oracle import fp8_oracle as F8
from oracle import vl2_oracle as O
from tests.emu.backend import emulated_backend
from tests.util import rel


@pytest.fixture(scope="module")
def emu():
    with emulated_backend() as lib:
        yield lib


def bf(*shape, scale=1.0, seed=0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed + sum(shape))) * scale).bfloat16()


def test_e4m3fn_codes_three_implementations_agree():
    """All 256 codes: the specification restated in python == PyTorch's float8_e4m3fn; encode(decode(c)) == c; the known-answer table; RNE
    ties and saturation through both encoders on a dense sweep."""
    codes = torch.arange(256, dtype=torch.uint8)
    tv = codes.view(torch.float8_e4m3fn).float()
    for c in range(256):
        v = F8.e4m3fn_decode(c)
        if math.isnan(v):
            assert c in (0x7F, 0xFF) and math.isnan(tv[c].item())
        else:
            assert tv[c].item() == v, c
            assert F8.e4m3fn_encode(v) == (c if c != 0x80 else 0x80), c
    for v, code in F8.KNOWN_E4M3FN:
        assert F8.e4m3fn_encode(v) == code, (v, hex(code))
        assert torch.tensor([v]).to(torch.float8_e4m3fn).view(torch.uint8).item() == code, (v, hex(code))
    # every midpoint and a point either side of it, both signs, up to the saturation boundary
    vals = []
    for c in range(0x7E):
        a, b = F8.e4m3fn_decode(c), F8.e4m3fn_decode(c + 1)
        vals += [a, (a + b) / 2, a + (b - a) * 0.49, a + (b - a) * 0.51]
    vals += [448.0, 455.9, 463.9]
    t = torch.tensor(vals + [-v for v in vals], dtype=torch.float32)
    tq = t.to(torch.float8_e4m3fn).view(torch.uint8)
    for v, q in zip(t.tolist(), tq.tolist()):
        assert F8.e4m3fn_encode(v) == q, (v, hex(q))


def test_row_scale_is_the_smallest_power_of_two(emu):
    amax = torch.tensor([0.0, 1e-45, 448.0, 448.0001, 447.9, 1.0, 0.02, 3.5, 7.0, 7.0001, 1e-3, 2.0 ** -20, 100.0, 896.0, 1e30])
    e = F8.row_scale_exponent(amax)
    for a, ee in zip(amax.tolist(), e.tolist()):
        if a < 2.0 ** -126:
            assert ee == 0
        elif abs(ee) < 100:
            assert a <= 448.0 * 2.0 ** ee and a > 448.0 * 2.0 ** (ee - 1), (a, ee)


def test_emu_quantiser_matches_the_oracle_bit_for_bit(emu):
    from videollama2_amd import ops
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(37, 256, generator=g) * 0.02).bfloat16()
    w[3] = 0                                                     # a zero row
    w[5] *= 1e-6                                                 # tiny rows, huge rows, a row whose maximum is exactly 448 * 2^k
    w[6] *= 1e6
    w[7, :] = torch.linspace(-7.0, 7.0, 256).bfloat16()
    w[8, 0] = 448.0
    w[9, 0] = -450.0                                             # bf16(450) = 450: just past 448 -> the next power of two
    q, sc = ops.quant_fp8(w)
    qo, so = F8.quant_rows(w)
    assert torch.equal(sc, so), (sc, so)
    assert torch.equal(q, qo), int((q != qo).sum())
    assert float(sc[3]) == 1.0 and int(q[3].max()) == 0
    assert float(sc[8]) == 1.0 and float(sc[9]) == 2.0
    d = F8.dequant(q, sc)                                        # a weight and its fp8 image differ by at most half a step: 2^-4 relative
    big = w.float().abs() > (w.float().abs().amax(dim=1, keepdim=True) * 2.0 ** -8)
    assert ((d - w.float()).abs()[big] <= w.float().abs()[big] * 2.0 ** -4 + 1e-30).all()


@pytest.mark.parametrize("case", ["plain", "bias_res", "f32_norm", "swiglu_rms", "two_pass"])
def test_emu_gemv_fp8_matches_the_oracle(emu, case):
    from videollama2_amd import ops
    N, K = (128, 256) if case != "two_pass" else (8, 16384)       # two_pass: more than 512 vectors of 16 weights per row
    w = bf(N, K, scale=K ** -0.5, seed=1)
    x = bf(K, seed=2)
    q, sc = ops.quant_fp8(w)
    if case in ("plain", "two_pass"):
        y = ops.gemv_fp8(q, sc, x, out_f32=True)
        ref = F8.gemv(q, sc, x)
    elif case == "bias_res":
        bias, res = torch.randn(N), bf(N, seed=3)
        y = ops.gemv_fp8(q, sc, x, bias=bias, res=res).float()
        ref = F8.gemv(q, sc, x, bias=bias, res=res)
    elif case == "f32_norm":
        nw = torch.rand(K) + 0.5
        y = ops.gemv_fp8(q, sc, x, norm_w=nw, eps=1e-5, out_f32=True)
        ref = F8.gemv(q, sc, x, norm_w=nw, eps=1e-5)
    else:
        y = ops.gemv_fp8(q, sc, x, eps=1e-5, swiglu=True, rms_plain=True).float()
        ref = F8.gemv(q, sc, x, eps=1e-5, swiglu=True, rms=True)
    tol = 1e-5 if y.dtype == torch.float32 and case in ("plain", "two_pass", "f32_norm") else 4e-3      # 16-bit outputs: one rounding
    assert rel(y, ref) < tol, rel(y, ref)
    # the same numbers through the 16-bit GEMV on the DEQUANTISED weights (exactly representable in bf16): isolates the kernel from the format
    wd = F8.dequant(q, sc).bfloat16()
    assert torch.equal(wd.float(), F8.dequant(q, sc))
    if case == "plain":
        assert rel(y, ops.gemv(wd, x, out_f32=True)) < 1e-5


def test_emu_fp8_decode_step_stage_equals_operators_and_tracks_dequantised_weights(emu, golden_small):
    """decoder.enable_fp8_decode(): (1) the stage-level call (vl2_llm_decode_step with VL2_STAGE_DECODE_FP8) == the per-operator loop, bit for
    bit; (2) the logits equal those of a decoder whose 16-bit weights were REPLACED by the dequantised fp8 copies (same arithmetic up to the
    fp32 summation order); (3) the format's own error against the unquantised decoder is what e4m3fn's 3 mantissa bits give (reported)."""
    from videollama2_amd import ops
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small
    cfg = g["cfg"]
    m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"]), "cpu", max_seq_len=64)
    dec = m.decoder
    base = dec.prefill(g["inputs_embeds"]).clone()
    pos0 = dec.pos
    kc, vc = [k.clone() for k in dec.kcache], [v.clone() for v in dec.vcache]

    def run(stage, n=3):
        for k, v, k0, v0 in zip(dec.kcache, dec.vcache, kc, vc):
            k.copy_(k0); v.copy_(v0)
        dec.pos = pos0
        dec.logits.copy_(base)
        dec.state.copy_(torch.tensor([pos0 - 1, 0], dtype=torch.int32))
        out = []
        for _ in range(n):
            if stage:
                d, _, ws = dec._stage_desc()
                ops.llm_decode_step(d, dec.logits, dec.tok, dec.state, dec.hist, dec.partial, ws, fp8=getattr(dec, "decode_fp8", False))
            else:
                ops.argmax(dec.logits, dec.tok, dec.hist, 0, dec.state)
                dec._decode_kernels(dyn=True)
            out.append((int(dec.tok), dec.logits.clone()))
        return out

    ref16 = run(True)
    dec.enable_fp8_decode()
    a, b = run(True), run(False)
    assert [t for t, _ in a] == [t for t, _ in b]
    for (_, la), (_, lb) in zip(a, b):
        assert torch.equal(la, lb)
    # (2) the dequantised weights through the 16-bit kernels
    dec.enable_fp8_decode(False)
    saved = [{k: lw[k].clone() for k in ("wqkv", "wo", "wgu", "wd")} for lw in dec.w["layers"]], dec.w["lm_head"].clone()
    for lw, q8 in zip(dec.w["layers"], dec.w8["layers"]):
        for k in ("wqkv", "wo", "wgu", "wd"):
            lw[k].copy_(F8.dequant(*q8[k]).to(lw[k].dtype))
    dec.w["lm_head"].copy_(F8.dequant(*dec.w8["lm_head"]).to(dec.w["lm_head"].dtype))
    c = run(True)
    for lw, s0 in zip(dec.w["layers"], saved[0]):
        for k in s0:
            lw[k].copy_(s0[k])
    dec.w["lm_head"].copy_(saved[1])
    e_kernel = max(rel(la, lc) for (_, la), (_, lc) in zip(a, c))
    e_format = rel(a[0][1], ref16[0][1])                         # step 1 feeds the same token in both; later steps may follow different tokens
    print(f"[fp8] emulator decode: fp8 kernels vs 16-bit kernels on the dequantised weights {e_kernel:.2e}; fp8 vs unquantised weights {e_format:.2e}")
    assert e_kernel < 2e-2                                       # 16-bit intermediate roundings may fall either way; the logits stay together
    assert e_format < 0.1
