"""The LAB build (videollama2_amd/libvl2hip_lab.so = the product sources with -DVL2_LAB, scripts/build_lab_lib.sh): the experiments that were
measured and lost stay buildable and under test without riding in the product library.  Every test here is skipped when the lab library has
not been built; the product library refuses the lab forms (checked below)."""
import pytest
import torch

from tests.util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
LAB_VARIANTS = (2, 5, 9, 10, 17, 18, 19, 20, 21, 22, 23, 25, 27, 28, 29, 62, 193, 225)


def bf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).bfloat16()


@pytest.fixture()
def ops():
    """videollama2_amd.ops routed through the lab library for the duration of one test."""
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from videollama2_amd import _lib, ops as o
    if not _lib.lab_built():
        pytest.skip("libvl2hip_lab.so not built (scripts/build_lab_lib.sh)")
    _lib.set_lab(True)
    _lib.load()
    try:
        yield o
    finally:
        o.set_gemm_variant(0)
        o.set_stage_flags(0)
        _lib.set_lab(False)


def test_product_library_refuses_the_lab_forms():
    from videollama2_amd import _lib, ops as o
    assert not _lib.lab()
    lib = _lib.load()
    assert not any(hasattr(lib, n) for n in _lib.LAB_SIGNATURES), "lab entry points exported by the product library"
    a, w = bf(256, 256).to(DEV), bf(256, 256).to(DEV)
    try:
        for v in LAB_VARIANTS:
            o.set_gemm_variant(v)
            with pytest.raises(_lib.Vl2HipError):
                o.gemm(a, w)
    finally:
        o.set_gemm_variant(0)
    with pytest.raises(_lib.Vl2HipError):
        o.attn_decode_fused(a, a, a, a, a, a, a, 32, 8, a, 1.0, a)


@pytest.mark.parametrize("M,N,K", [(300, 512, 448), (9232, 4096, 1024), (1621, 28672, 4096)])
def test_lab_gemm_variants_are_bit_identical(ops, M, N, K):
    """5 = the 128x256 kernel with the woven LDS-DMA issue, 9 = gemm8 (four waves, 128x128 wave tiles), 62 = the persistent 192-row form with two
    accumulator sets, 193 / 225 = the fill-the-round tiles with the woven issue; VL2_GEMM_WEAVE4 / NO_WEAVE4 = the other issue order of the
    256- / 192-row tiles: all the family's bits."""
    a, w, bias, res = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV), torch.randn(N).to(DEV), bf(M, N).to(DEV)
    ops.set_gemm_variant(1)
    ref = (ops.gemm(a, w, bias=bias, res=res, act=ops.ACT_QGELU), ops.gemm(a, w, bias=bias, act=ops.ACT_QGELU), ops.gemm(a, w))
    run = lambda: (ops.gemm(a, w, bias=bias, res=res, act=ops.ACT_QGELU), ops.gemm(a, w, bias=bias, act=ops.ACT_QGELU), ops.gemm(a, w))
    for v in (5, 9, 10, 193, 225, 62):                          # 10 = the 160-row tile of round 6 (the 192-row tile without group 1's third row block)
        ops.set_gemm_variant(v)
        for _ in range(2):
            assert all(torch.equal(x, y) for x, y in zip(run(), ref)), (M, N, K, v)
    for v, fl in ((8, ops.STAGE_WEAVE4), (12, ops.STAGE_NO_WEAVE4), (0, ops.STAGE_WEAVE), (0, ops.STAGE_WEAVE4)):
        ops.set_gemm_variant(v)
        ops.set_stage_flags(fl)
        assert all(torch.equal(x, y) for x, y in zip(run(), ref)), (M, N, K, v, fl)
        ops.set_stage_flags(0)


@pytest.mark.parametrize("M,N,K", [(1621, 4096, 4096), (700, 512, 1024), (300, 1024, 14336), (1621, 28672, 1024)])      # (the last: a row-split call = the mixed launch, without SwiGLU)
def test_lab_mfma16_set_is_one_arithmetic(ops, M, N, K):
    """Every tile of the 16 x 16 x 32 set -- the 256 x 256 ping-pong tile (variants 16 / 26 = 32- / 64-deep phases; a row-split call = the mixed launch), the
    one-round 128 x 128 body (256 + VL2_GEMM_MFMA16) and the fill-the-round tiles (224 / 192 + the flag: k_gemm7.h gemm7_loop16, lab) -- accumulates a dot
    product as the same sequence of 32-product steps: the same bits ON THE GPU, with residual + row statistics + the producer-side finalize."""
    a, w, bias, res = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV), torch.randn(N).to(DEV), bf(M, N).to(DEV)
    ref = None
    for v in (16, 26, 256, 224, 192, 0):
        ops.set_gemm_variant(v)
        st, rn, tick = torch.zeros(M, N // 64, 2, device=DEV), torch.zeros(M, 2, device=DEV), torch.zeros(M // 64 + 2, dtype=torch.int32, device=DEV)
        y = ops.gemm(a, w, bias=bias, res=res, stats_out=st, norm_out=(ops.NORM_RMS, 1e-6, rn, tick), mfma16=True)
        y2 = ops.gemm(a, w, mfma16=True)
        if ref is None:
            ref = (y, st, rn, y2)
            assert rel(y, a.float() @ w.float().T + bias + res.float()) < 1e-2
        else:
            assert torch.equal(y, ref[0]) and torch.equal(st, ref[1]) and torch.equal(rn, ref[2]) and torch.equal(y2, ref[3]), (M, N, K, v)
        assert int(tick.abs().sum().item()) == 0, v


def test_lab_gemm9_issue_orders_equal_variant_16(ops):
    """csrc/k_gemm9.h MODE 1 ... 6, 9 (variants 17 ... 22, 26): other orders of the same sums on the 16 x 16 x 32 instruction -> variant 16's bits."""
    M, N, K = 1621, 4096, 4096
    a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
    ops.set_gemm_variant(16)
    ref = ops.gemm(a, w)
    for v in (17, 18, 19, 20, 21, 22, 26):
        ops.set_gemm_variant(v)
        assert torch.equal(ops.gemm(a, w), ref), v


def test_lab_gate_up_group_depth_and_tail_switch_equal_the_shipped_launch(ops):
    """Round 6: the mixed 16 x 16 x 32 launch of gate/up with other depths of the row-tile group that shares a W panel on an XCD (27 / 29 = 8 / 6) and with the
    empty waves of the ragged tail tiles computing (28: the launch as it was before the skip) -> the shipped launch's bits (variant 26), ragged and whole M."""
    from videollama2_amd.weights import pack_gate_up
    for M in (1621, 945, 1536 + 128):
        a = bf(M, 4096).to(DEV)
        wgu = pack_gate_up(bf(2048, 4096, scale=1 / 64, seed=3), bf(2048, 4096, scale=1 / 64, seed=4)).to(DEV)
        rn = ops.row_norm_finalize(ops.row_stats(a), 4096, ops.NORM_RMS, 1e-6)
        ops.set_gemm_variant(26)
        ref = ops.gemm(a, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None))
        for v in (27, 28, 29):
            ops.set_gemm_variant(v)
            assert torch.equal(ops.gemm(a, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None)), ref), (M, v)
    ops.set_gemm_variant(0)


def test_lab_causal_attention_without_the_hidden_tile_skip(ops):
    """vl2_attn_fwd variant 5 (lab): the one-stream causal kernel computing the tiles the mask hides from a whole wave -> variant 3's bits (every P of such a tile is 0)."""
    D, smax = 128, 2048
    for S, nh, nkv, off in ((1621, 32, 8, 0), (200, 4, 2, 0), (300, 8, 2, 700)):
        q, kc, vc = bf(S, nh * D).to(DEV), bf(nkv, smax, D).to(DEV), bf(nkv, smax, D, seed=1).to(DEV)
        outs = {}
        try:
            for v in (3, 5):
                ops.set_attn_kv_groups(v)
                outs[v] = torch.zeros(S, nh * D, dtype=torch.bfloat16, device=DEV)
                ops.attn_fwd(q, kc, vc, outs[v], (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, S + off, nh // nkv, D ** -0.5, True, off, D)
        finally:
            ops.set_attn_kv_groups(0)
        assert torch.equal(outs[3], outs[5]), (S, nh, nkv, off)


def test_lab_gemm_stream_k_variant(ops):
    """Stream-K form (tuning knob 2): partial tiles cross workgroups through the caller-owned workspace with an
    agent-scope release/acquire hand-off; repeated launches screen for stale reads.  fp32 sums in a different order."""
    M, N, K = 1621, 4096, 4096
    a, w, res = bf(M, K), bf(N, K, scale=K ** -0.5), bf(M, N)
    ad, wd, rd = a.to(DEV), w.to(DEV), res.to(DEV)
    ref = ops.gemm(ad, wd, res=rd).float()
    ops.attach_workspace(DEV)
    try:
        ops.set_gemm_variant(2)
        for _ in range(5):
            out = ops.gemm(ad, wd, res=rd).float()
            assert (out - ref).abs().max().item() <= 0.04 * ref.abs().max().item()
            assert rel(out, ref) < 2e-3
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("nh,nkv,smax,pos", [(32, 8, 2048, 300), (32, 8, 2048, 1650), (28, 4, 512, 300), (32, 8, 2048, 63), (32, 8, 2048, 64),
                                             (32, 8, 512, 511), (64, 8, 1024, 700)])
def test_lab_attn_decode_fused_combine_equals_two_kernels(ops, nh, nkv, smax, pos):
    """vl2_attn_decode_fused (the kv head's last-finishing slice combines its q heads inside the attention launch; what
    vl2_llm_decode_step enqueues) must give the bits of vl2_attn_decode (attention + combine kernels): same output, same appended
    cache rows, every ticket counter at the number of live slices -- Mistral (group 4), Qwen2-7B (group 7 = two head blocks per kv
    head), 72B (group 8), slice boundaries and the last cache row, repeated launches."""
    HD = 128
    qkv, kc, vc = bf((nh + 2 * nkv) * HD, seed=pos), bf(nkv, smax, HD, seed=2), bf(nkv, smax, HD, seed=3)
    inv = 1.0 / (1e6 ** (torch.arange(0, HD, 2).float() / HD))
    fr = torch.arange(smax).float()[:, None] * inv[None]
    cos_t, sin_t = fr.cos().contiguous().to(DEV), fr.sin().contiguous().to(DEV)
    nsp, group = (smax + 63) // 64, nh // nkv
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    for rep in range(3):
        k1, v1, k2, v2 = kc.to(DEV), vc.to(DEV), kc.to(DEV), vc.to(DEV)
        p1, p2 = torch.full((nh * nsp * 130,), 7.0, device=DEV), torch.full((nh * nsp * 130,), -3.0, device=DEV)   # stale garbage
        o1, o2 = torch.zeros(nh * HD, dtype=torch.bfloat16, device=DEV), torch.ones(nh * HD, dtype=torch.bfloat16, device=DEV)
        ops.attn_decode(qkv.to(DEV), k1, v1, cos_t, sin_t, p1, o1, nh, nkv, pos, HD ** -0.5, pos_dev=pos_dev, ctx_cap=smax)
        cnt = torch.zeros(nkv, dtype=torch.int32, device=DEV)
        ops.attn_decode_fused(qkv.to(DEV), k2, v2, cos_t, sin_t, p2, o2, nh, nkv, pos_dev, HD ** -0.5, cnt)
        assert torch.equal(o1, o2), (rep, int((o1 != o2).sum()))
        assert torch.equal(k1, k2) and torch.equal(v1, v2)
        assert cnt.tolist() == [((pos + 64) // 64) * ((group + 3) // 4)] * nkv




def test_lab_decode_step_forms_equal_the_default(ops):
    """VL2_STAGE_FUSED_DECODE_ATTN (attention + elected combine in one launch) and VL2_STAGE_DECODE_TAIL (o_proj / gate-up / down as one persistent
    launch with two grid barriers): the default decode step's bits, at Mistral-7B widths on two layers."""
    from oracle import vl2_oracle as O
    from videollama2_amd.decoder import HipMistralDecoder
    cfg = O.config_videollama2_7b(16)
    cfg["llm"]["num_hidden_layers"] = 2
    keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
    sd = O.seeded_state_dict(cfg, 5, only=keep)
    x = (torch.randn(70, 4096, generator=torch.Generator().manual_seed(3)).bfloat16().float() * 0.5).to(DEV)
    outs = []
    for fl in (0, ops.STAGE_FUSED_DECODE_ATTN, ops.STAGE_DECODE_TAIL):
        ops.set_stage_flags(fl)
        dec = HipMistralDecoder(cfg, sd, DEV, max_seq_len=512)
        toks, logits = dec.generate(x, max_new_tokens=6, return_logits=True, use_graph=True)
        outs.append((toks.cpu(), logits.cpu()))
        ops.set_stage_flags(0)
    for t, l in outs[1:]:
        assert torch.equal(t, outs[0][0]) and torch.equal(l, outs[0][1])
