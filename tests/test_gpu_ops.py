"""Per-operator parity on MI355X: every libvl2hip.so kernel, called through the C ABI (videollama2_amd.ops ->
ctypes), against the fp32 oracle of the same op on identical bf16-rounded inputs.  Tolerances: tests/util.py."""

import pytest
import torch
import torch.nn.functional as F

from tests.util import TOL_BF16_OUT, TOL_F32_OUT, rel

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


@pytest.mark.parametrize("M,N,K", [(200, 256, 192), (128, 128, 64), (9232, 1024, 1024), (1621, 6144, 4096), (577, 4096, 1024)])
def test_gemm_plain_f32(ops, M, N, K):
    a, w = bf(M, K), bf(N, K, scale=K ** -0.5)
    c = ops.gemm(a.to(DEV), w.to(DEV), out_f32=True)
    assert rel(c, a.float() @ w.float().T) < TOL_F32_OUT


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_bias_act_res(ops, act):
    M, N, K = 1521, 1024, 4096
    a, w, bias, res = bf(M, K), bf(N, K, scale=K ** -0.5), torch.randn(N), bf(M, N)
    c = ops.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), res=res.to(DEV), act=act)
    y = F.linear(a.float(), w.float(), bias)
    y = [y, y * torch.sigmoid(1.702 * y), F.gelu(y), F.silu(y)][act] + res.float()
    assert rel(c, y) < TOL_BF16_OUT


@pytest.mark.parametrize("M,N,K,kind,variants", [(9232, 1024, 1024, 2, (0, 1, 8, 12)), (9232, 1024, 4096, 2, (0, 12)), (1621, 4096, 4096, 1, (0, 1, 4, 224, 192, 256)),
                                                 (1621, 4096, 14336, 1, (0, 4, 224)), (945, 4096, 4096, 1, (0, 1)), (333, 4096, 4096, 1, (0, 32))])
def test_gemm_producer_side_finalize_equals_the_launch(ops, M, N, K, kind, variants):
    """k_gemm.h gemm_rows_ticket (round 5): the statistics-producing GEMMs of the step (ViT out_proj / fc2, decoder o / down at S = 1621 and at the
    T = 8 length, a small-M shape that falls back to the appended launch) leave (mean, rstd) of their output rows in `row_norm_out` -- bit for bit
    what vl2_row_norm_finalize computes from `stats_out`, on every tile shape, repeatedly on ONE ticket block (the kernels re-arm it), racing
    workgroups on eight XCDs included (write-through partials, agent-scope loads in the electing workgroup)."""
    a, w, bias, res = bf(M, K), bf(N, K, scale=K ** -0.5), torch.randn(N), bf(M, N)
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    eps = 1e-5
    tick = torch.zeros(M // 64 + 2, dtype=torch.int32, device=DEV)
    try:
        ops.set_gemm_variant(1)
        st_ref = torch.zeros(M, N // 64, 2, device=DEV)
        y_ref = ops.gemm(ad, wd, bias=bd, res=rd, stats_out=st_ref)
        rn_ref = ops.row_norm_finalize(st_ref, N, kind, eps)
        for v in variants:
            ops.set_gemm_variant(v)
            for rep in range(4):
                st, rn = torch.zeros(M, N // 64, 2, device=DEV), torch.full((M, 2), -7.0, device=DEV)
                y = ops.gemm(ad, wd, bias=bd, res=rd, stats_out=st, norm_out=(kind, eps, rn, tick))
                assert torch.equal(y, y_ref) and torch.equal(st, st_ref), (v, rep)
                assert torch.equal(rn, rn_ref), (v, rep, float((rn - rn_ref).abs().max()))
                assert int(tick.abs().sum().item()) == 0, f"variant {v}: tickets not re-armed"
        ops.set_gemm_variant(0)
        ops.set_stage_flags(ops.STAGE_NO_TICKET_OPS)                       # the launch appended by vl2_gemm (A/B form)
        st, rn = torch.zeros(M, N // 64, 2, device=DEV), torch.zeros(M, 2, device=DEV)
        ops.gemm(ad, wd, bias=bd, res=rd, stats_out=st, norm_out=(kind, eps, rn, tick))
        assert torch.equal(rn, rn_ref)
    finally:
        ops.set_gemm_variant(0)
        ops.set_stage_flags(0)


@pytest.mark.parametrize("M,N,K", [(300, 512, 448), (9232, 4096, 1024), (1621, 28672, 4096)])
def test_gemm_pingpong_variant_is_bit_identical(ops, M, N, K):
    """The ping-pong kernels (128x256: knob 4, 256x256: knob 8, 192x256: knob 12) accumulate in the same order as the 128x128 kernel:
    results must be bit-identical (also screens the counted-vmcnt / phase-barrier schedules for races)."""
    a, w, bias, res = bf(M, K), bf(N, K, scale=K ** -0.5), torch.randn(N), bf(M, N)
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    try:
        ops.set_gemm_variant(1)
        ref = ops.gemm(ad, wd, bias=bd, res=rd, act=ops.ACT_QGELU)
        for v in (4, 8, 12, 0):                                      # (the lab forms 5 / 9 / 10: tests/test_gpu_lab.py)
            ops.set_gemm_variant(v)
            for _ in range(3):
                assert torch.equal(ops.gemm(ad, wd, bias=bd, res=rd, act=ops.ACT_QGELU), ref)
        ops.set_gemm_variant(1)
        ref = ops.gemm(ad, wd, bias=bd, act=ops.ACT_QGELU)           # no residual: the register-resident C^T epilogue of 4 / 8 / 12
        for v in (4, 8, 12, 0):
            ops.set_gemm_variant(v)
            assert torch.equal(ops.gemm(ad, wd, bias=bd, act=ops.ACT_QGELU), ref)
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(1621, 4096, 14336), (1621, 4096, 4096), (1521, 4096, 4096), (1400, 4096, 1024), (225, 128, 64), (3000, 1024, 2048)])
def test_gemm_fill_round_kernel_is_bit_identical(ops, M, N, K):
    """csrc/k_gemm7.h (224 x 128 / 192 x 128 tiles, knobs 224 / 192; the automatic choice for the decoder's o / down projections at
    S = 1621 and the connector's GEMMs on 1521 positions): same K order and epilogue arithmetic as the 128 x 128 kernel -> the same bits,
    with every epilogue those callers use (residual + row statistics; bias + GELU; plain; fp32; the norm-carrying forms).  Repeated
    launches screen the counted-vmcnt ring (5 / 6 pieces per wave) and the shared-image epilogue for races."""
    a, w, bias, res = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV), torch.randn(N).to(DEV), bf(M, N).to(DEV)
    xs = bf(M, K, seed=3).to(DEV)
    st = ops.row_stats(xs)
    colsum = w.float().sum(1)

    def run():
        so = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=DEV)
        outs = (ops.gemm(a, w, res=res, stats_out=so), so, ops.gemm(a, w, bias=bias, act=ops.ACT_GELU), ops.gemm(a, w), ops.gemm(a, w, out_f32=True))
        if K % 128 == 0:
            outs += (ops.gemm(xs, w, bias=bias, norm=(ops.NORM_LN, st, 1e-5, colsum)), ops.gemm(xs, w, norm=(ops.NORM_RMS, st, 1e-5, None)))
        return outs
    try:
        ops.set_gemm_variant(1)
        ref = run()
        for v in (224, 192, 0):                                       # (225 / 193 = the woven issue: lab, tests/test_gpu_lab.py)
            ops.set_gemm_variant(v)
            for _ in range(3):
                assert all(torch.equal(x, y) for x, y in zip(run(), ref)), (M, N, K, v)
    finally:
        ops.set_gemm_variant(0)


def test_gemm_fill_round_gathered_is_bit_identical(ops):
    """The Conv3d(k2, s2, p1) taps as a gathered GEMM on the fill-the-round tiles (T = 16: 1521 x 4096 x 32768, the automatic choice) and on
    a small grid with every border case, against the 128 x 128 gathered kernel."""
    from videollama2_amd.connector import conv3d_k2s2p1_index
    for T, H, C, N in ((16, 24, 4096, 4096), (4, 6, 1024, 256)):
        x, wp, b = bf(T * H * H, C).to(DEV), bf(N, 8 * C, scale=(8 * C) ** -0.5).to(DEV), torch.randn(N, device=DEV)
        idx, _ = conv3d_k2s2p1_index(T, H, H, DEV)
        zero = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
        try:
            ops.set_gemm_variant(1)
            ref = ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C))
            for v in (192, 224, 0):
                ops.set_gemm_variant(v)
                for _ in range(2):
                    assert torch.equal(ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C)), ref), (T, v)
        finally:
            ops.set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(1621, 28672, 4096), (1100, 16384, 2048)])
def test_gemm_row_split_swiglu_is_bit_identical(ops, M, N, K):
    """Wide-N GEMMs whose M leaves the 256-row kernel a nearly empty last row tile run as two launches (256-row kernel on the
    first floor(M/256)*256 rows, the chooser's pick on the rest): same bits as the 128x128 kernel on the whole matrix."""
    a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
    try:
        ops.set_gemm_variant(1)
        ref = ops.gemm(a, w, swiglu=True)
        ops.set_gemm_variant(0)
        for _ in range(3):
            assert torch.equal(ops.gemm(a, w, swiglu=True), ref)
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(9216, 4096, 4096), (4608, 4096, 2048)])
def test_gemm_round_split_is_bit_identical(ops, M, N, K):
    """Whole rounds + a short tail (the STC s1 convolutions: 576 big tiles = 2.25 rounds): the rows of the whole rounds run on the
    256x256 kernel, the tail rows on the chooser's pick for them -- same bits as one kernel on the whole matrix, with every
    epilogue the connector uses (plain, +bias +GELU, +residual and row statistics)."""
    a, w, res, bias = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV), bf(M, N).to(DEV), torch.randn(N).to(DEV)
    st = [torch.zeros((M, N // 64, 2), dtype=torch.float32, device=DEV) for _ in range(2)]
    try:
        ops.set_gemm_variant(8)
        ref = (ops.gemm(a, w), ops.gemm(a, w, bias=bias, act=ops.ACT_GELU), ops.gemm(a, w, res=res, stats_out=st[0]))
        ops.set_gemm_variant(0)
        for _ in range(2):
            got = (ops.gemm(a, w), ops.gemm(a, w, bias=bias, act=ops.ACT_GELU), ops.gemm(a, w, res=res, stats_out=st[1]))
            assert all(torch.equal(x, y) for x, y in zip(got, ref)) and torch.equal(st[0], st[1])
    finally:
        ops.set_gemm_variant(0)


def test_gemm_small_m_kernel_bit_identical(ops):
    """Small-M shapes take the 64x64 kernel automatically; it must give the bits of the 128x128 kernel (knob 1)."""
    from videollama2_amd.connector import conv3d_k2s2p1_index
    for M, N, K, kw in [(1154, 1024, 4096, dict(bias=True, res=True)), (338, 4096, 4096, dict()), (169, 4096, 4096, dict(act=ops.ACT_GELU, bias=True)),
                        (1154, 1024, 1024, dict(bias=True, res=True)), (77, 256, 64, dict(act=ops.ACT_QGELU, bias=True))]:
        a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
        bias = torch.randn(N, device=DEV) if kw.get("bias") else None
        res = bf(M, N).to(DEV) if kw.get("res") else None
        out = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
        try:
            ops.set_gemm_variant(32)
            forced = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
            ops.set_gemm_variant(1)
            ref = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
        finally:
            ops.set_gemm_variant(0)
        assert torch.equal(forced, ref) and torch.equal(out, ref), (M, N, K)
    T, H, C = 4, 24, 1024
    x, wp, b = bf(T * H * H, C).to(DEV), bf(C, 8 * C, scale=(8 * C) ** -0.5).to(DEV), torch.randn(C, device=DEV)
    idx, (To, Ho, Wo) = conv3d_k2s2p1_index(T, H, H, DEV, to_range=(1, 3))
    zero = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
    out = ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C))
    try:
        ops.set_gemm_variant(1)
        ref = ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C))
    finally:
        ops.set_gemm_variant(0)
    assert torch.equal(out, ref)


def test_gemm_splitk_small_grids(ops):
    """Opt-in: small-grid GEMMs split K once a workspace is attached (include/vl2hip.h VL2_TUNE_SPLITK): the per-rank shapes of the
    frame-sharded encoder and a long-K gathered Conv3d.  Same products, fp32 partial sums in a different order; the tile
    counters are re-armed by the kernel, so repeated launches need no host reset and must be bit-stable."""
    from videollama2_amd.connector import conv3d_k2s2p1_index
    ops.attach_workspace(DEV)
    cases = [(1154, 1024, 4096, dict(bias=True, res=True)), (338, 4096, 4096, dict()), (169, 4096, 4096, dict(act=ops.ACT_GELU, bias=True)),
             (507, 4096, 2048, dict(bias=True, res=True))]
    for M, N, K, kw in cases:
        a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
        bias = torch.randn(N, device=DEV) if kw.get("bias") else None
        res = bf(M, N).to(DEV) if kw.get("res") else None
        ref = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0)).float()
        try:
            ops.set_splitk(True)
            first = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
            assert rel(first, ref) < 2e-3 and (first.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item()
            for _ in range(4):
                assert torch.equal(ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0)), first)
        finally:
            ops.set_splitk(False)
    T, H, C = 4, 24, 1024                                       # 2 output frames... K = 8192: 128 K-tiles, 3 x 8 tiles
    x, wp, b = bf(T * H * H, C).to(DEV), bf(C, 8 * C, scale=(8 * C) ** -0.5).to(DEV), torch.randn(C, device=DEV)
    idx, (To, Ho, Wo) = conv3d_k2s2p1_index(T, H, H, DEV, to_range=(1, 3))
    zero = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
    ref = ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C)).float()
    try:
        ops.set_splitk(True)
        out = ops.gemm(x, wp, bias=b, act=3, gather=(idx, zero, C))
    finally:
        ops.set_splitk(False)
    assert out.shape == (2 * Ho * Wo, C) and rel(out, ref) < 2e-3


def test_gemm_swiglu(ops):
    from videollama2_amd.weights import pack_gate_up
    M, I, K = 333, 1792, 1024
    a, wg, wu = bf(M, K), bf(I, K, scale=K ** -0.5), bf(I, K, scale=K ** -0.5, seed=1)
    c = ops.gemm(a.to(DEV), pack_gate_up(wg, wu).to(DEV), swiglu=True)
    assert c.shape == (M, I)
    assert rel(c, F.silu(a.float() @ wg.float().T) * (a.float() @ wu.float().T)) < TOL_BF16_OUT


def test_gemm_conv3d_gather_full_size(ops):
    """Conv3d(4096,4096,k2,s2,p1)+bias+SiLU on [4,24,24] frames as the gathered GEMM (projector.py:164-174, :208)."""
    from videollama2_amd.connector import conv3d_k2s2p1_index
    T, H, C = 4, 24, 512
    x = bf(T * H * H, C)
    w3 = bf(C, C, 2, 2, 2, scale=(8 * C) ** -0.5)
    bias = torch.randn(C) * 0.1
    idx, (To, Ho, Wo) = conv3d_k2s2p1_index(T, H, H, DEV)
    wp = w3.permute(0, 2, 3, 4, 1).reshape(C, 8 * C).contiguous()
    y = ops.gemm(x.to(DEV), wp.to(DEV), bias=bias.to(DEV), act=3, gather=(idx, torch.zeros(C, dtype=torch.bfloat16, device=DEV), C))
    ref = F.silu(F.conv3d(x.float().view(1, T, H, H, C).permute(0, 4, 1, 2, 3), w3.float(), bias, stride=2, padding=1))
    ref = ref[0].permute(1, 2, 3, 0).reshape(To * Ho * Wo, C)
    assert (To, Ho, Wo) == (3, 13, 13)
    assert rel(y, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("C,rows", [(128, 37), (1024, 9232), (1152, 1458), (4096, 1521), (3584, 301), (8192, 333), (8192, 1)])
def test_layernorm_rmsnorm(ops, C, rows):
    x, w, b, r = bf(rows, C, scale=2.0) + 0.5, torch.randn(C), torch.randn(C), bf(rows, C)
    x = x.bfloat16()
    y = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, res=r.to(DEV), silu=True)
    assert rel(y, F.silu(F.layer_norm(x.float(), (C,), w, b, 1e-5) + r.float())) < TOL_BF16_OUT
    y = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert rel(y, F.layer_norm(x.float(), (C,), w, b, 1e-5)) < TOL_BF16_OUT
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
    xf = x.float()
    assert rel(y, w * xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)) < TOL_BF16_OUT


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_patch_embed_full_size(ops, dtype):
    """[T,3,336,336] -> cat([cls, conv(k14,s14)]) + pos  (HF CLIPVisionEmbeddings.forward)."""
    T, D, P, S = 2, 1024, 14, 336
    fr = torch.randn(T, 3, S, S, generator=torch.Generator().manual_seed(3)).to(dtype)
    w = bf(D, 3, P, P, scale=588 ** -0.5)
    pos, cls = bf(577, D, scale=0.1), bf(D, scale=0.5)
    a = ops.patchify(fr.to(DEV), P, 640)
    pw = torch.zeros(D, 640, dtype=torch.bfloat16)
    pw[:, :588] = w.reshape(D, 588)
    x = torch.zeros(T * 577, D, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, pw.to(DEV), res=pos.to(DEV), out=x, out_map=(576, 1, 1), res_map=(576, 1))
    ops.fill_cls(x, (cls.float() + pos[0].float()).bfloat16().to(DEV), T, 577)
    ref = F.conv2d(fr.bfloat16().float(), w.float(), stride=P).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.float().expand(T, 1, D), ref], 1) + pos.float()[None]
    assert rel(x.view(T, 577, D), ref) < TOL_BF16_OUT


def test_attn_vit_full_size(ops):
    """B frames x 16 heads x 577 tokens x 64, non-causal, straight out of the fused qkv buffer."""
    B, H, N, D = 2, 16, 577, 64
    qkv = bf(B * N, 3 * H * D)
    o = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device=DEV)
    g = qkv.to(DEV)
    st = (N * 3 * H * D, D, 3 * H * D)
    ops.attn_fwd(g, g[:, H * D:], g[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
    q, k, v = [t.view(B, N, H, D).transpose(1, 2).float() for t in qkv.view(B * N, 3, H * D).unbind(1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert rel(o, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("groups", [0, 1, 2])       # KV groups per workgroup: auto / one / two (split KV tiles, LDS merge)
@pytest.mark.parametrize("S,nh,nkv", [(1621, 32, 8), (200, 4, 2), (64, 2, 1), (1345, 8, 2)])
def test_attn_causal_gqa(ops, S, nh, nkv, groups):
    D, smax = 128, 2048
    q, kc, vc = bf(S, nh * D), bf(nkv, smax, D), bf(nkv, smax, D, seed=1)
    o = torch.zeros(S, nh * D, dtype=torch.bfloat16, device=DEV)
    ops.set_attn_kv_groups(groups)
    try:
        ops.attn_fwd(q.to(DEV), kc.to(DEV), vc.to(DEV), o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D),
                     1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D)
    finally:
        ops.set_attn_kv_groups(0)
    qf = q.view(S, nh, D).transpose(0, 1).float()
    kf = kc[:, :S].float().repeat_interleave(nh // nkv, 0)
    vf = vc[:, :S].float().repeat_interleave(nh // nkv, 0)
    sc = (qf @ kf.transpose(1, 2)) * D ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vf).transpose(0, 1).reshape(S, nh * D)
    assert rel(o, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("scale", [0.05, 0.0713, 0.0884, 0.131])
def test_attn_causal_two_groups_rows_with_only_masked_keys(ops, scale):
    """Second KV group: rows above the diagonal of its first tile meet only masked keys there; their partial result must carry
    weight 0 for any rounding of (-1e30 * scale)."""
    S, nh, nkv, D, smax = 330, 4, 2, 128, 512
    q, kc, vc = bf(S, nh * D, seed=3), bf(nkv, smax, D, seed=4), bf(nkv, smax, D, seed=5)
    o = torch.zeros(S, nh * D, dtype=torch.bfloat16, device=DEV)
    ops.set_attn_kv_groups(2)
    try:
        ops.attn_fwd(q.to(DEV), kc.to(DEV), vc.to(DEV), o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D),
                     1, nh, S, S, nh // nkv, scale, True, 0, D)
    finally:
        ops.set_attn_kv_groups(0)
    qf = q.view(S, nh, D).transpose(0, 1).float()
    kf, vf = kc[:, :S].float().repeat_interleave(2, 0), vc[:, :S].float().repeat_interleave(2, 0)
    sc = (qf @ kf.transpose(1, 2) * scale).masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vf).transpose(0, 1).reshape(S, nh * D)
    assert torch.isfinite(o.float()).all() and rel(o, ref) < TOL_BF16_OUT


def test_attn_softmax_spike(ops):
    """One key dominating one query row at a late tile forces the online-softmax rescale branch (guide rule 26)."""
    B, H, N, D = 1, 1, 300, 64
    qkv = bf(B * N, 3 * D)
    qkv[17, :D] = 6.0
    qkv[250, D:2 * D] = 6.0
    g = qkv.to(DEV)
    o = torch.zeros(N, D, dtype=torch.bfloat16, device=DEV)
    st = (N * 3 * D, D, 3 * D)
    ops.attn_fwd(g, g[:, D:], g[:, 2 * D:], o, st, st, st, (N * D, D, D), B, H, N, N, 1, D ** -0.5, False, 0, D)
    q, k, v = [t.float() for t in qkv.view(N, 3, D).unbind(1)]
    ref = torch.softmax(q @ k.T * D ** -0.5, -1) @ v
    assert rel(o, ref) < TOL_BF16_OUT
    assert rel(o[17], ref[17]) < 1e-2


@pytest.mark.parametrize("Fr,H,W,C", [(2, 13, 13, 4096), (1, 5, 7, 1024), (1, 2, 1, 256), (2, 3, 18, 1024), (1, 24, 24, 8192)])
def test_dwconv_odd_grids(ops, Fr, H, W, C):
    """The 13x13 grid of STC stage s2 at full width, a non-square grid and a single column."""
    x = bf(Fr * H * W, C)
    wt = (torch.randn(C, 1, 3, 3) * 0.3).bfloat16().float()
    lnw, lnb = torch.randn(C), torch.randn(C)
    y = ops.dwconv3x3_ln_silu(x.to(DEV), wt.view(C, 9).t().contiguous().to(DEV), lnw.to(DEV), lnb.to(DEV), Fr, H, W)
    ref = F.conv2d(x.float().view(Fr, H, W, C).permute(0, 3, 1, 2), wt, padding=1, groups=C).permute(0, 2, 3, 1)
    ref = F.silu(F.layer_norm(ref, (C,), lnw, lnb, 1e-5)).reshape(Fr * H * W, C)
    assert rel(y, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("C", [4096, 8192])          # 8192: the STC connector in front of the 72B decoder
def test_stc_direct_kernels_full_width(ops, C):
    Fr, H = 2, 24
    x = bf(Fr * H * H, C)
    wt, lnw, lnb = torch.randn(C, 1, 3, 3) * 0.3, torch.randn(C), torch.randn(C)
    wt = wt.bfloat16().float()
    y = ops.dwconv3x3_ln_silu(x.to(DEV), wt.view(C, 9).t().contiguous().to(DEV), lnw.to(DEV), lnb.to(DEV), Fr, H, H)
    ref = F.conv2d(x.float().view(Fr, H, H, C).permute(0, 3, 1, 2), wt, padding=1, groups=C).permute(0, 2, 3, 1)
    ref = F.silu(F.layer_norm(ref, (C,), lnw, lnb, 1e-5)).reshape(Fr * H * H, C)
    assert rel(y, ref) < TOL_BF16_OUT
    m = ops.chan_mean(x.to(DEV), Fr, H * H)
    assert rel(m, x.float().view(Fr, H * H, C).mean(1)) < 1e-5
    w1, b1 = bf(C // 4, C, scale=C ** -0.5), torch.randn(C // 4) * 0.1
    w2, b2 = bf(C, C // 4, scale=1 / 32), torch.randn(C) * 0.1
    g1 = ops.small_linear(m, w1.to(DEV), b1.to(DEV), ops.ACT_SILU)
    g2 = ops.small_linear(g1, w2.to(DEV), b2.to(DEV), ops.ACT_SIGMOID)
    mref = x.float().view(Fr, H * H, C).mean(1)
    gref = torch.sigmoid(F.linear(F.silu(F.linear(mref, w1.float(), b1)), w2.float(), b2))
    assert rel(g2, gref) < 1e-4
    xs = x.to(DEV).clone()
    ops.se_scale_(xs, g2, Fr, H * H)
    assert rel(xs.view(Fr, H * H, C), x.float().view(Fr, H * H, C) * gref[:, None, :]) < TOL_BF16_OUT


@pytest.mark.parametrize("Fr,H,W,C,rd", [(3, 24, 24, 4096, 1024), (2, 13, 13, 4096, 1024), (1, 5, 7, 1024, 256), (1, 2, 1, 256, 64), (2, 3, 18, 1024, 48),
                                         (2, 27, 27, 3584, 896), (1, 24, 24, 8192, 2048)])
def test_dwconv_strip_squeeze_and_fused_excite(ops, Fr, H, W, C, rd):
    """Round 4: the strip form of the depthwise kernel (taps in LDS, persistent teams) with the SE squeeze folded in, and the one-launch
    excite + scale, against the per-position kernels and small_linear + se_scale (timm SEModule as restated in oracle/shims/timm)."""
    x = bf(Fr * H * W, C).to(DEV)
    wt = (torch.randn(C, 1, 3, 3) * 0.3).bfloat16().float()
    lnw, lnb = torch.randn(C).to(DEV), torch.randn(C).to(DEV)
    w9c = wt.view(C, 9).t().contiguous().to(DEV)
    y0 = ops.dwconv3x3_ln_silu(x, w9c, lnw, lnb, Fr, H, W)
    y, m = ops.dwconv3x3_ln_silu_mean(x, w9c, lnw, lnb, Fr, H, W)
    if W >= 16:
        assert torch.equal(y, y0)                     # same fmaf chain as the four-position kernel
    else:
        assert rel(y, y0) < 2e-3
    ref = F.conv2d(x.float().cpu().view(Fr, H, W, C).permute(0, 3, 1, 2), wt, padding=1, groups=C).permute(0, 2, 3, 1)
    ref = F.silu(F.layer_norm(ref, (C,), lnw.cpu(), lnb.cpu(), 1e-5)).reshape(Fr * H * W, C)
    assert rel(y, ref) < TOL_BF16_OUT
    assert rel(m, y.float().view(Fr, H * W, C).mean(1)) < 1e-5
    if Fr > 1:                                        # a frame's bits do not depend on how many frames the launch holds
        y1, m1 = ops.dwconv3x3_ln_silu_mean(x[H * W:2 * H * W].contiguous(), w9c, lnw, lnb, 1, H, W)
        assert torch.equal(y1, y[H * W:2 * H * W]) and torch.equal(m1, m[1:2])
    w1, b1 = bf(rd, C, scale=C ** -0.5).to(DEV), (torch.randn(rd) * 0.1).to(DEV)
    w2, b2 = bf(C, rd, scale=rd ** -0.5).to(DEV), (torch.randn(C) * 0.1).to(DEV)
    g1 = ops.small_linear(m, w1, b1, ops.ACT_SILU)
    ya = ops.se_scale_(y.clone(), ops.small_linear(g1, w2, b2, ops.ACT_SIGMOID), Fr, H * W)
    yb = ops.se_excite_scale_(y.clone(), g1, w2, b2, Fr, H * W)
    gate = torch.sigmoid(F.linear(g1, w2.float(), b2))
    assert rel(yb.view(Fr, H * W, C), y.float().view(Fr, H * W, C) * gate[:, None, :]) < TOL_BF16_OUT
    assert rel(yb, ya) < 1e-3 and (yb != ya).float().mean() < 0.02


def test_rope_kv(ops):
    S, nh, nkv, HD, smax, pos0 = 77, 32, 8, 128, 256, 100
    qkv = bf(S, (nh + 2 * nkv) * HD)
    inv = 1.0 / (1e6 ** (torch.arange(0, HD, 2).float() / HD))
    fr = torch.arange(smax).float()[:, None] * inv[None]
    q_out = torch.zeros(S, nh * HD, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(nkv, smax, HD, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    ops.rope_kv(qkv.to(DEV), q_out, kc, vc, fr.cos().contiguous().to(DEV), fr.sin().contiguous().to(DEV), nh, nkv, pos0)
    rot = lambda t: torch.cat([-t[..., HD // 2:], t[..., :HD // 2]], -1)
    emb = torch.cat([fr, fr], -1)[pos0:pos0 + S]
    c, s = emb.cos()[:, None], emb.sin()[:, None]
    q = qkv[:, :nh * HD].float().view(S, nh, HD)
    k = qkv[:, nh * HD:(nh + nkv) * HD].float().view(S, nkv, HD)
    v = qkv[:, (nh + nkv) * HD:].view(S, nkv, HD)
    assert rel(q_out.view(S, nh, HD), q * c + rot(q) * s) < TOL_BF16_OUT
    assert rel(kc[:, pos0:pos0 + S].transpose(0, 1), k * c + rot(k) * s) < TOL_BF16_OUT
    assert torch.equal(vc[:, pos0:pos0 + S].transpose(0, 1).cpu(), v)
    assert kc[:, :pos0].abs().max().item() == 0 and kc[:, pos0 + S:].abs().max().item() == 0


@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (32000, 4096), (40, 256)])
def test_gemv(ops, N, K):
    w, x, nw, res = bf(N, K, scale=K ** -0.5), bf(K), torch.randn(K), bf(N)
    y = ops.gemv(w.to(DEV), x.to(DEV), res=res.to(DEV))
    assert rel(y, w.float() @ x.float() + res.float()) < TOL_BF16_OUT
    yf = ops.gemv(w.to(DEV), x.to(DEV), norm_w=nw.to(DEV), eps=1e-5, out_f32=True)
    xf = x.float()
    xn = (nw * xf * torch.rsqrt(xf.pow(2).mean() + 1e-5)).bfloat16().float()
    assert rel(yf, w.float() @ xn) < TOL_F32_OUT


def test_gemv_swiglu(ops):
    from videollama2_amd.weights import pack_gate_up
    I, K = 14336, 4096
    wg, wu, x = bf(I, K, scale=K ** -0.5), bf(I, K, scale=K ** -0.5, seed=1), bf(K)
    y = ops.gemv(pack_gate_up(wg, wu).to(DEV), x.to(DEV), swiglu=True)
    assert rel(y, F.silu(wg.float() @ x.float()) * (wu.float() @ x.float())) < TOL_BF16_OUT


@pytest.mark.parametrize("pos", [0, 62, 63, 64, 1699])
def test_attn_decode_fused_rope_append(ops, pos):
    """One new token at position `pos`: RoPE(q,k) + cache append + attention over rows [0, pos], vs the fp32 oracle."""
    nh, nkv, smax, HD = 32, 8, 2048, 128
    qkv, kc, vc = bf((nh + 2 * nkv) * HD), bf(nkv, smax, HD), bf(nkv, smax, HD, seed=1)
    inv = 1.0 / (1e6 ** (torch.arange(0, HD, 2).float() / HD))
    fr = torch.arange(smax).float()[:, None] * inv[None]
    cos_t, sin_t = fr.cos().contiguous(), fr.sin().contiguous()
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    part = torch.zeros(nh * ((smax + 63) // 64) * 130, device=DEV)
    out = torch.zeros(nh * HD, dtype=torch.bfloat16, device=DEV)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    for dyn in (False, True):
        kcd.copy_(kc); vcd.copy_(vc)
        ops.attn_decode(qkv.to(DEV), kcd, vcd, cos_t.to(DEV), sin_t.to(DEV), part, out, nh, nkv, pos, HD ** -0.5,
                        pos_dev=pos_dev if dyn else None, ctx_cap=smax if dyn else 0)
        rot = lambda t: torch.cat([-t[..., HD // 2:], t[..., :HD // 2]], -1)
        emb = torch.cat([fr[pos], fr[pos]])
        q = qkv[:nh * HD].float().view(nh, HD)
        kn = qkv[nh * HD:(nh + nkv) * HD].float().view(nkv, HD)
        vn = qkv[(nh + nkv) * HD:].view(nkv, HD)
        qr = (q * emb.cos() + rot(q) * emb.sin()).bfloat16().float()
        kr = (kn * emb.cos() + rot(kn) * emb.sin()).bfloat16()
        assert torch.equal(kcd[:, pos].cpu(), kr) and torch.equal(vcd[:, pos].cpu(), vn)           # appended rows
        assert torch.equal(kcd[:, :pos].cpu(), kc[:, :pos]) and torch.equal(kcd[:, pos + 1:].cpu(), kc[:, pos + 1:])
        kf = torch.cat([kc[:, :pos], kr[:, None]], 1).float().repeat_interleave(nh // nkv, 0)
        vf = torch.cat([vc[:, :pos], vn[:, None]], 1).float().repeat_interleave(nh // nkv, 0)
        a = torch.softmax(torch.einsum("hd,hkd->hk", qr, kf) * HD ** -0.5, -1)
        assert rel(out, torch.einsum("hk,hkd->hd", a, vf).reshape(-1)) < TOL_BF16_OUT


def test_argmax_embed(ops):
    lg = torch.randn(32000)
    lg[1234] = 9.0
    lg[31999] = 9.0          # tie -> first index, like torch.argmax
    tok = torch.zeros(1, dtype=torch.int32, device=DEV)
    hist = torch.zeros(4, dtype=torch.int32, device=DEV)
    ops.argmax(lg.to(DEV), tok, hist, 2)
    assert tok.item() == 1234 and hist.tolist() == [0, 0, 1234, 0]
    state = torch.tensor([10, 1], dtype=torch.int32, device=DEV)           # device-side {position, step} (graph replay form)
    ops.argmax(lg.to(DEV), tok, hist, 0, state)
    assert hist.tolist() == [0, 1234, 1234, 0] and state.tolist() == [11, 2]
    ids = torch.tensor([3, 0, 31999], dtype=torch.int32, device=DEV)
    tab = bf(32000, 256)
    out = torch.zeros(3, 256, dtype=torch.bfloat16, device=DEV)
    ops.embed_rows(ids, tab.to(DEV), out)
    assert torch.equal(out.cpu(), tab[ids.cpu().long()])


def test_abi_rejects_bad_shapes(ops):
    from videollama2_amd._lib import Vl2HipError
    a, w = bf(16, 100).to(DEV), bf(128, 100).to(DEV)
    with pytest.raises(Vl2HipError):
        ops.gemm(a, w)                      # K % 64 != 0
    with pytest.raises(Vl2HipError):
        ops.gemm(bf(16, 64).to(DEV), bf(100, 64).to(DEV))   # N % 128 != 0


def test_small_linear_is_independent_of_the_frame_slot(ops):
    """A frame's SE vector must come out bit-identical whichever slot of the 8-frame pass it occupies (= however many frames
    the rank holds): the frame-sharded encoder relies on it.  (Regression: -ffast-math re-associated the unrolled FMA chains
    differently per slot.)"""
    g = torch.Generator().manual_seed(0)
    for N, K in ((1024, 4096), (4096, 1024), (96, 384)):
        x = torch.randn(11, K, generator=g).to(DEV)
        w, b = bf(N, K, scale=K ** -0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
        full = ops.small_linear(x, w, b, ops.ACT_SILU)
        for lo, hi in ((0, 2), (3, 8), (5, 11), (10, 11)):
            assert torch.equal(ops.small_linear(x[lo:hi].contiguous(), w, b, ops.ACT_SILU), full[lo:hi]), (N, K, lo, hi)


def test_attn_fwd_head_dim_96_padded_72(ops):
    """SigLIP-so400m attention: head_dim 72 zero-padded to 96, 729 tokens, scale 72^-0.5, non-causal."""
    B, H, N, D, hd = 2, 16, 729, 96, 72
    qkv = bf(B * N, 3 * H * D)
    qkv.view(B * N, 3 * H, D)[:, :, hd:] = 0
    qd = qkv.to(DEV)
    o = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device=DEV)
    st = (N * 3 * H * D, D, 3 * H * D)
    ops.attn_fwd(qd, qd[:, H * D:], qd[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, hd ** -0.5, False, 0, D)
    q, k, v = [t.view(B, N, H, D).transpose(1, 2).float() for t in qkv.view(B * N, 3, H * D).unbind(1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert rel(o, ref) < TOL_BF16_OUT
    assert o.view(B * N, H, D)[:, :, hd:].abs().max().item() == 0


@pytest.mark.parametrize("S,mean,std", [(336, (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)), (384, (0.5,) * 3, (0.5,) * 3)])
def test_patchify_u8_matches_normalise_then_patchify(ops, S, mean, std):
    """uint8 ingest: patch rows from raw uint8 [T,H,W,3] frames (normalise in registers) vs the processor's fp32 arithmetic
    followed by the float patchify: identical except on bf16 rounding ties (reciprocal instead of division)."""
    T, P = 3, 14
    u8 = torch.randint(0, 256, (T, S, S, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(S))
    x = u8.float() * (1 / 255)
    frames = ((x - torch.tensor(mean)) / torch.tensor(std)).permute(0, 3, 1, 2).contiguous()
    kp = (3 * P * P + 63) // 64 * 64
    rows_f = ops.patchify(frames.to(DEV), P, kp)
    rows_u = ops.patchify_u8(u8.to(DEV), P, kp, 1 / 255, mean, std)
    assert rows_u.shape == rows_f.shape == (T * (S // P) ** 2, kp)
    same = (rows_f == rows_u).float().mean().item()
    ulp = ((rows_f.float() - rows_u.float()).abs() / rows_f.float().abs().clamp_min(1e-3)).max().item()
    assert same > 0.995 and ulp <= 2 ** -7, (same, ulp)
    assert rows_u[:, 3 * P * P:].abs().max().item() == 0


def test_gemv_batched_rows_equal_single_row(ops):
    """Multi-row GEMV at 7B shapes: each of the MB rows equals the single-row kernel bit for bit, incl. the K = 14336 case
    that only fits 2 rows of x in LDS per pass (the launcher splits a batch of 4)."""
    from videollama2_amd.weights import pack_gate_up
    for N, K, kw in ((6144, 4096, dict(norm=True)), (4096, 14336, dict(res=True)), (4096, 4096, dict(res=True, bias=True))):
        w = bf(N, K, scale=K ** -0.5).to(DEV)
        x = bf(4, K).to(DEV)
        nw = (1 + 0.1 * torch.randn(K)).to(DEV) if kw.get("norm") else None
        res = bf(4, N).to(DEV) if kw.get("res") else None
        bias = torch.randn(N).to(DEV) if kw.get("bias") else None
        for mb in (2, 3, 4):
            got = ops.gemv_batched(w, x[:mb], norm_w=nw, res=None if res is None else res[:mb], bias=bias)
            for b in range(mb):
                one = ops.gemv(w, x[b], norm_w=nw, res=None if res is None else res[b], bias=bias)
                assert torch.equal(got[b], one), (N, K, mb, b)
    wgu = pack_gate_up(bf(2048, 4096, scale=1 / 64), bf(2048, 4096, scale=1 / 64, seed=1)).to(DEV)
    x, nw = bf(4, 4096).to(DEV), (1 + 0.1 * torch.randn(4096)).to(DEV)
    got = ops.gemv_batched(wgu, x, norm_w=nw, swiglu=True)
    for b in range(4):
        assert torch.equal(got[b], ops.gemv(wgu, x[b], norm_w=nw, swiglu=True))


def test_gemm_skinny_7b_shapes(ops):
    """Skinny-M GEMM at the decode shapes of a batched step (M = 5..64), every epilogue, vs torch and vs the tiled GEMM."""
    from videollama2_amd.weights import pack_gate_up
    ops.attach_workspace(DEV)
    for M, N, K, kw in ((8, 6144, 4096, dict(bias=True)), (16, 4096, 4096, dict(res=True)), (64, 4096, 14336, dict(res=True)),
                        (33, 32000, 4096, dict(f32=True)), (5, 4608, 3584, dict(bias=True))):
        a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
        bias = torch.randn(N).to(DEV) if kw.get("bias") else None
        res = bf(M, N).to(DEV) if kw.get("res") else None
        got = ops.gemm_skinny(a, w, bias=bias, res=res, out_f32=bool(kw.get("f32")))
        ref = a.float() @ w.float().T + (bias if bias is not None else 0) + (res.float() if res is not None else 0)
        assert rel(got, ref) < (TOL_F32_OUT if kw.get("f32") else TOL_BF16_OUT), (M, N, K)
        assert rel(got, ops.gemm(a, w, bias=bias, res=res, out_f32=bool(kw.get("f32"))).float()) < 3e-3
        for _ in range(2):
            assert torch.equal(ops.gemm_skinny(a, w, bias=bias, res=res, out_f32=bool(kw.get("f32"))), got)   # deterministic
    a = bf(24, 4096).to(DEV)
    wg, wu = bf(2048, 4096, scale=1 / 64), bf(2048, 4096, scale=1 / 64, seed=1)
    got = ops.gemm_skinny(a, pack_gate_up(wg, wu).to(DEV), swiglu=True)
    ref = F.silu(a.float().cpu() @ wg.float().T) * (a.float().cpu() @ wu.float().T)
    assert rel(got, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("M,N,K", [(300, 512, 448), (1621, 28672, 4096), (2500, 1024, 14336)])
def test_gemm_mfma16_kernel(ops, M, N, K):
    """csrc/k_gemm9.h (variant 16 / VL2_GEMM_MFMA16 / STAGE_MFMA16): the 256 x 256 ping-pong tile on v_mfma_f32_16x16x32_bf16.  The one kernel that is NOT
    bit-identical with the family (the instruction sums 32 products per accumulation step, the family's 16): held to the fp32 result at the
    family's tolerance, to the family within two bf16 output roundings, deterministic, and a row's bits do not depend on M.  The decoder's
    gate/up shape (SwiGLU + RMSNorm carried) and a residual-carrying projection."""
    from videollama2_amd.weights import fold_norm, pack_gate_up
    a, w, bias, res = bf(M, K), bf(N, K, scale=K ** -0.5), torch.randn(N), bf(M, N)
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    g = (1 + 0.3 * torch.randn(K, generator=torch.Generator().manual_seed(5))).bfloat16().float()
    wg, wu = bf(N // 2, K, scale=K ** -0.5, seed=3), bf(N // 2, K, scale=K ** -0.5, seed=4)
    wgu, _, _ = fold_norm(pack_gate_up(wg, wu), g, dev=DEV)
    rn = ops.row_norm_finalize(ops.row_stats(ad), K, ops.NORM_RMS, 1e-6)
    af = ad.float()
    h = af * torch.rsqrt(af.pow(2).mean(-1, keepdim=True) + 1e-6) * g.to(DEV)
    ref = F.linear(af, wd.float(), bd) + rd.float()
    ref_sw = F.silu(h @ wg.to(DEV).float().T) * (h @ wu.to(DEV).float().T)
    fam = (ops.gemm(ad, wd, bias=bd, res=rd), ops.gemm(ad, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None)))
    try:
        ops.set_gemm_variant(16)
        y, y_sw = ops.gemm(ad, wd, bias=bd, res=rd), ops.gemm(ad, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None))
        assert rel(y, ref) < TOL_BF16_OUT and rel(y_sw, ref_sw) < TOL_BF16_OUT, (rel(y, ref), rel(y_sw, ref_sw))
        assert rel(y, fam[0]) < 2 * TOL_BF16_OUT and rel(y_sw, fam[1]) < 2 * TOL_BF16_OUT
        assert rel(y, ref) < 1.1 * rel(fam[0], ref) + 1e-5 and rel(y_sw, ref_sw) < 1.1 * rel(fam[1], ref_sw) + 1e-5     # as accurate as the family
        for _ in range(3):
            assert torch.equal(ops.gemm(ad, wd, bias=bd, res=rd), y) and torch.equal(ops.gemm(ad, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None)), y_sw)
        m1 = M - 130
        assert torch.equal(ops.gemm(ad[:m1], wd, bias=bd, res=rd[:m1]), y[:m1])
        assert torch.equal(ops.gemm(ad[:m1], wgu, swiglu=True, norm=(ops.NORM_RMS, rn[:m1], 1e-6, None)), y_sw[:m1])
        ops.set_gemm_variant(0)
        ops.set_stage_flags(ops.STAGE_MFMA16)                       # the session switch reaches SwiGLU calls only
        assert torch.equal(ops.gemm(ad, wgu, swiglu=True, norm=(ops.NORM_RMS, rn, 1e-6, None)), y_sw) and torch.equal(ops.gemm(ad, wd, bias=bd, res=rd), fam[0])
    finally:
        ops.set_gemm_variant(0)
        ops.set_stage_flags(0)


def test_gemm_one_round_kernel_bit_identical(ops):
    """Grids of <= 256 tiles with K >= 4096 take the deep-ring 128x128 kernel automatically; same bits as the 2-stage kernel."""
    for M, N, K, kw in [(845, 4096, 4096, dict()), (945, 4096, 14336, dict(res=True)), (2308, 1024, 4096, dict(bias=True, res=True, act=ops.ACT_GELU))]:
        a, w = bf(M, K).to(DEV), bf(N, K, scale=K ** -0.5).to(DEV)
        bias = torch.randn(N, device=DEV) if kw.get("bias") else None
        res = bf(M, N).to(DEV) if kw.get("res") else None
        out = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
        try:
            ops.set_gemm_variant(256)
            forced = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
            ops.set_gemm_variant(1)
            ref = ops.gemm(a, w, bias=bias, res=res, act=kw.get("act", 0))
        finally:
            ops.set_gemm_variant(0)
        assert torch.equal(forced, ref) and torch.equal(out, ref), (M, N, K)


@pytest.mark.parametrize("M,N,K,kind,swiglu", [(9232, 3072, 1024, 2, False),       # ViT layer_norm1 -> q/k/v (256x256 kernel)
                                               (9232, 4096, 1024, 2, False),       # ViT layer_norm2 -> fc1 + QuickGELU
                                               (1621, 6144, 4096, 1, False),       # Mistral input_layernorm -> q/k/v
                                               (1621, 28672, 4096, 1, True),       # post_attention_layernorm -> gate/up + SwiGLU (row split)
                                               (300, 512, 384, 2, False)])         # small / ragged
def test_gemm_norm_carrying_chain_full_width(ops, M, N, K, kind, swiglu):
    """Producer GEMM (residual fused) emits row statistics == vl2_row_stats of its output, bit for bit; the consumer GEMM
    normalises in its epilogue (weights.fold_norm) and matches fp32 norm -> linear within one bf16 output rounding; every
    kernel variant reproduces the auto choice's bits."""
    from videollama2_amd.weights import fold_norm, pack_gate_up
    a0, w0, res = bf(M, 256, seed=9), bf(K, 256, scale=1 / 16, seed=10), bf(M, K, scale=2.0, seed=11)
    st = torch.zeros((M, K // 64, 2), device=DEV)
    x = ops.gemm(a0.to(DEV), w0.to(DEV), res=res.to(DEV), stats_out=st)          # the residual stream [M, K]
    assert torch.equal(st, ops.row_stats(x))
    xf = x.float().cpu()
    g, b = (1 + 0.3 * torch.randn(K)).bfloat16().float(), (0.2 * torch.randn(K)).bfloat16().float()
    eps = 1e-5
    h = F.layer_norm(xf, (K,), g, b, eps) if kind == 2 else xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * g
    if swiglu:
        wg, wu = bf(N // 2, K, scale=K ** -0.5, seed=12), bf(N // 2, K, scale=K ** -0.5, seed=13)
        wp, _, _ = fold_norm(pack_gate_up(wg, wu), g, dev=DEV)
        run = lambda: ops.gemm(x, wp, swiglu=True, norm=(ops.NORM_RMS, st, eps, None))
        ref = F.silu(h @ wg.float().T) * (h @ wu.float().T)
    else:
        w, c = bf(N, K, scale=K ** -0.5, seed=12), torch.randn(N).bfloat16().float()
        wp, s, t = fold_norm(w, g, b if kind == 2 else None, c, dev=DEV)
        bias = t if kind == 2 else c.to(DEV)
        act = ops.ACT_QGELU if N == 4096 else ops.ACT_NONE
        run = lambda: ops.gemm(x, wp, bias=bias, act=act, norm=(kind, st, eps, s if kind == 2 else None))
        ref = F.linear(h, w.float(), c)
        ref = ref * torch.sigmoid(1.702 * ref) if act else ref
    y = run()
    assert rel(y, ref) < TOL_BF16_OUT
    try:
        for v in (1, 4, 8, 12):
            ops.set_gemm_variant(v)
            assert torch.equal(run(), y), f"variant {v}"
    finally:
        ops.set_gemm_variant(0)


def test_gemm_operands_beyond_2gib_run_in_chunks(ops):
    """VideoLLaMA2-72B's lm_head [152064 x 8192] bf16 is 2.49 GB: beyond the kernels' 32-bit buffer offsets.  vl2_gemm covers it
    in column chunks; rows of W below, across and above the 2 GiB line are checked against fp32 matmuls of the same slices."""
    N, K, M = 152064, 8192, 96
    g = torch.Generator(device=DEV).manual_seed(3)
    w = (torch.randn((N, K), generator=g, device=DEV, dtype=torch.float32) * K ** -0.5).to(torch.bfloat16)
    a = bf(M, K, seed=4).to(DEV)
    y = ops.gemm(a, w, out_f32=True)
    split = (2 ** 31 - 65536) // 2 // K                                    # first row of W whose bytes start past the 32-bit range
    for r0 in (0, split - 300, split - 40, split + 512, N - 128):
        ref = a.float() @ w[r0:r0 + 128].float().T
        assert rel(y[:, r0:r0 + 128], ref) < TOL_F32_OUT, r0
    yb = ops.gemm(a, w)                                                    # bf16 output, auto kernel choice
    assert rel(yb[:, split - 64:split + 64], y[:, split - 64:split + 64]) < TOL_BF16_OUT


@pytest.mark.parametrize("B,H,N,D", [(2, 16, 577, 64), (1, 2, 64, 64), (3, 4, 130, 128)])
def test_attn_second_structure_noncausal(ops, B, H, N, D):
    """csrc/k_attn2.h (variant 3): LDS-DMA K/V ring + transpose-read V, against torch and against the first structure."""
    qkv = bf(B * N, 3 * H * D, seed=N)
    g = qkv.to(DEV)
    st = (N * 3 * H * D, D, 3 * H * D)
    outs = {}
    try:
        for var in (1, 3):
            ops.set_attn_kv_groups(var)
            o = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device=DEV)
            for _ in range(3):                                          # repeated launches: screens the DMA / barrier schedule for races
                ops.attn_fwd(g, g[:, H * D:], g[:, 2 * H * D:], o, st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
            outs[var] = o
    finally:
        ops.set_attn_kv_groups(0)
    q, k, v = [t.view(B, N, H, D).transpose(1, 2).float() for t in qkv.view(B * N, 3, H * D).unbind(1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert rel(outs[3], ref) < TOL_BF16_OUT and rel(outs[3], outs[1]) < 3e-3


@pytest.mark.parametrize("S,nh,nkv", [(1621, 32, 8), (200, 4, 2), (64, 2, 1), (1345, 8, 2), (2973, 32, 8)])
def test_attn_second_structure_causal_gqa(ops, S, nh, nkv):
    D, smax = 128, 4096
    q, kc, vc = bf(S, nh * D), bf(nkv, smax, D), bf(nkv, smax, D, seed=1)
    o = torch.zeros(S, nh * D, dtype=torch.bfloat16, device=DEV)
    ops.set_attn_kv_groups(3)
    try:
        for _ in range(3):
            ops.attn_fwd(q.to(DEV), kc.to(DEV), vc.to(DEV), o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D),
                         1, nh, S, S, nh // nkv, D ** -0.5, True, 0, D)
    finally:
        ops.set_attn_kv_groups(0)
    qf = q.view(S, nh, D).transpose(0, 1).float()
    kf = kc[:, :S].float().repeat_interleave(nh // nkv, 0)
    vf = vc[:, :S].float().repeat_interleave(nh // nkv, 0)
    sc = (qf @ kf.transpose(1, 2)) * D ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vf).transpose(0, 1).reshape(S, nh * D)
    assert rel(o, ref) < TOL_BF16_OUT


@pytest.mark.parametrize("S,nh,nkv,off", [(945, 32, 8, 0), (1621, 32, 8, 0), (200, 4, 2, 0), (64, 2, 1, 0), (65, 2, 1, 0), (129, 4, 2, 0), (1345, 8, 2, 0), (300, 8, 2, 700)])
def test_attn_two_key_streams_causal_gqa(ops, S, nh, nkv, off):
    """k_attn2.h NS = 2 (variant 4; automatic for causal head_dim 128 while a sequence has <= 352 (query block, head) pairs): even / odd key tiles on two
    groups of four waves, merged through LDS.  1, 2, 3 tiles (the odd stream empty / shorter), a chunk of new rows against a longer cache, the workload's
    shapes; against torch, against the one-stream form (P is rounded relative to each stream's own running maximum: two 16-bit roundings apart at most),
    deterministic over repeated launches, and the automatic choice is a function of the SEQUENCE: a prompt's bits are the same alone and in a batch of 3."""
    D, smax, nk = 128, 4096, S + off
    q, kc, vc = bf(S, nh * D), bf(nkv, smax, D), bf(nkv, smax, D, seed=1)
    qd, kd, vd = q.to(DEV), kc.to(DEV), vc.to(DEV)
    outs = {}
    try:
        for var in (3, 4, 0):
            ops.set_attn_kv_groups(var)
            outs[var] = []
            for _ in range(3):
                o = torch.zeros(S, nh * D, dtype=torch.bfloat16, device=DEV)
                ops.attn_fwd(qd, kd, vd, o, (0, D, nh * D), (0, smax * D, D), (0, smax * D, D), (0, D, nh * D), 1, nh, S, nk, nh // nkv, D ** -0.5, True, off, D)
                outs[var].append(o)
        ops.set_attn_kv_groups(0)
        B = 3
        qb = torch.stack([qd, torch.zeros_like(qd), qd]).contiguous()
        kb, vb = torch.stack([kd, kd, kd]).contiguous(), torch.stack([vd, vd, vd]).contiguous()
        ob = torch.zeros(B, S, nh * D, dtype=torch.bfloat16, device=DEV)
        ops.attn_fwd(qb, kb, vb, ob, (S * nh * D, D, nh * D), (nkv * smax * D, smax * D, D), (nkv * smax * D, smax * D, D), (S * nh * D, D, nh * D),
                     B, nh, S, nk, nh // nkv, D ** -0.5, True, off, D)
    finally:
        ops.set_attn_kv_groups(0)
    qf = q.view(S, nh, D).transpose(0, 1).float()
    kf = kc[:, :nk].float().repeat_interleave(nh // nkv, 0)
    vf = vc[:, :nk].float().repeat_interleave(nh // nkv, 0)
    vis = torch.arange(nk)[None, :] <= torch.arange(S)[:, None] + off
    ref = (torch.softmax((qf @ kf.transpose(1, 2) * D ** -0.5).masked_fill(~vis, float("-inf")), -1) @ vf).transpose(0, 1).reshape(S, nh * D)
    assert rel(outs[4][0], ref) < TOL_BF16_OUT and rel(outs[4][0], outs[3][0]) < 3e-3
    assert all(torch.equal(o, outs[4][0]) for o in outs[4][1:])
    auto_is_two = ((S + 127) // 128) * nh <= 352
    assert torch.equal(outs[0][0], outs[4 if auto_is_two else 3][0])
    assert torch.equal(ob[0], outs[0][0]) and torch.equal(ob[2], outs[0][0])


def test_attn_two_key_streams_noncausal(ops):
    """The other instances of the two-stream form (not chosen automatically): ViT shape with the class-token peel, ragged head_dim 64, head_dim 128."""
    for B, H, N, D in ((2, 16, 577, 64), (2, 2, 150, 64), (1, 4, 130, 128)):
        qkv = bf(B * N, 3 * H * D, seed=N)
        g = qkv.to(DEV)
        st = (N * 3 * H * D, D, 3 * H * D)
        outs = {}
        try:
            for var in (3, 4):
                ops.set_attn_kv_groups(var)
                outs[var] = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device=DEV)
                for _ in range(3):
                    ops.attn_fwd(g, g[:, H * D:], g[:, 2 * H * D:], outs[var], st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
        finally:
            ops.set_attn_kv_groups(0)
        q, k, v = [t.view(B, N, H, D).transpose(1, 2).float() for t in qkv.view(B * N, 3, H * D).unbind(1)]
        ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, H * D)
        assert rel(outs[4], ref) < TOL_BF16_OUT and rel(outs[4], outs[3]) < 3e-3, (B, H, N, D)


def test_attn_class_token_peel_matches_plain_tiling(ops):
    """k_attn2.h CLS = true (automatic for the CLIP tower's 577 = 1 + 576 tokens): class key as the initial softmax state, class query in a
    dead wave of the last query block, nine key tiles -- against the plain tiling (variant 3) and torch at the workload's shape."""
    B, H, N, D = 4, 16, 577, 64
    g = bf(B * N, 3 * H * D, seed=3)
    g[7, :D] = 4.0
    g[0, H * D:H * D + D] = 4.0                                            # a query that loves the class key
    g = g.to(DEV)
    st = (N * 3 * H * D, D, 3 * H * D)
    outs = {}
    try:
        for var in (3, 0):
            ops.set_attn_kv_groups(var)
            outs[var] = torch.zeros(B * N, H * D, dtype=torch.bfloat16, device=DEV)
            for _ in range(3):
                ops.attn_fwd(g, g[:, H * D:], g[:, 2 * H * D:], outs[var], st, st, st, (N * H * D, D, H * D), B, H, N, N, 1, D ** -0.5, False, 0, D)
    finally:
        ops.set_attn_kv_groups(0)
    q, k, v = [t.view(B, N, H, D).transpose(1, 2).float() for t in g.cpu().view(B * N, 3, H * D).unbind(1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert rel(outs[0].cpu(), ref) < TOL_BF16_OUT and rel(outs[0], outs[3]) < 3e-3
    cls_rows = torch.arange(B) * N
    assert rel(outs[0].cpu()[cls_rows], ref[cls_rows]) < TOL_BF16_OUT


def test_attn_second_structure_softmax_spike(ops):
    B, H, N, D = 1, 1, 300, 64
    qkv = bf(B * N, 3 * D)
    qkv[17, :D] = 6.0
    qkv[250, D:2 * D] = 6.0
    g = qkv.to(DEV)
    o = torch.zeros(N, D, dtype=torch.bfloat16, device=DEV)
    st = (N * 3 * D, D, 3 * D)
    ops.set_attn_kv_groups(3)
    try:
        ops.attn_fwd(g, g[:, D:], g[:, 2 * D:], o, st, st, st, (N * D, D, D), B, H, N, N, 1, D ** -0.5, False, 0, D)
    finally:
        ops.set_attn_kv_groups(0)
    q, k, v = [t.float() for t in qkv.view(N, 3, D).unbind(1)]
    ref = torch.softmax(q @ k.T * D ** -0.5, -1) @ v
    assert rel(o, ref) < TOL_BF16_OUT and rel(o[17], ref[17]) < 1e-2


def test_pack_entry_points_match_weights_py_on_device(ops):
    """vl2_pack_* (one-time weight re-layout through the C ABI) against weights.py's tensor-op packing, on the device."""
    from tests.test_emu_pipeline import check_pack_entry_points_against_weights_py
    check_pack_entry_points_against_weights_py(ops, DEV)
    # full-width: Mistral gate/up and the layer-norm fold of a ViT q/k/v block
    from videollama2_amd import weights as Wt
    gate, up = bf(14336, 4096, scale=0.02).to(DEV), bf(14336, 4096, scale=0.02, seed=1).to(DEV)
    assert torch.equal(ops.pack_gate_up(gate, up), Wt.pack_gate_up(gate, up))
    w, g, b, c = bf(3072, 1024, scale=0.03).to(DEV), (1 + 0.1 * torch.randn(1024)).bfloat16().to(DEV), bf(1024, scale=0.1).to(DEV), bf(3072, scale=0.1).to(DEV)
    wp, s, t = ops.pack_fold_norm(w, g, b, c)
    wp0, s0, t0 = Wt.fold_norm(w, g, b, c, DEV)
    assert torch.equal(wp, wp0) and torch.allclose(s, s0, rtol=0, atol=1e-3) and torch.allclose(t, t0, rtol=0, atol=1e-3)
