"""Stage-level C-ABI entry points ON THE GPU (include/vl2hip.h vl2_vit_forward / vl2_stc_forward / vl2_llm_prefill /
vl2_llm_decode_step): one C call per stage, the layer loops inside libvl2hip.so.  They must give bit-identical results to the
per-operator host loops (same kernels, same order, same stream), for both families; the decode step must be capturable into a
hipGraph (it is what decoder.capture_graph captures) and the graph replay must equal eager calls of the same entry point."""
import pytest
import torch

from oracle import vl2_oracle as O
from tests.util import rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("family", ["v2", "v21"])
def test_stage_calls_equal_per_operator_path_on_device(golden_small, golden_small_v21, family):
    from videollama2_amd import ops
    from videollama2_amd.model import VideoLLaMA2Hip
    g = golden_small if family == "v2" else golden_small_v21
    cfg = g["cfg"]
    m = VideoLLaMA2Hip(cfg, O.seeded_state_dict(cfg, g["seed"]), DEV, max_seq_len=64)
    dec = m.decoder
    outs = {}
    try:
        for stage in (True, False):
            ops.STAGE_ABI = stage
            assert ops.stage_enabled() == stage
            tower = m.vision_tower(g["frames"].to(DEV))
            tower_u8 = m.vision_tower(g["frames_u8"].to(DEV))
            vis = m.mm_projector(tower.unsqueeze(0))
            logits = dec.prefill(g["inputs_embeds"].to(DEV)).clone()
            dec.state.copy_(torch.tensor([dec.pos - 1, 0], dtype=torch.int32))
            steps = []
            for _ in range(4):
                if stage:
                    d, _, ws = dec._stage_desc()
                    ops.llm_decode_step(d, dec.logits, dec.tok, dec.state, dec.hist, dec.partial, ws)
                else:
                    ops.argmax(dec.logits, dec.tok, dec.hist, 0, dec.state)
                    dec._decode_kernels(dyn=True)
                steps.append((int(dec.tok), dec.logits.clone()))
            outs[stage] = (tower, tower_u8, vis, logits, steps, dec.state.clone(), dec.hist[:4].clone())
    finally:
        ops.STAGE_ABI = True
    a, b = outs[True], outs[False]
    for i in (0, 1, 2, 3, 5, 6):
        assert torch.equal(a[i], b[i]), i
    assert [t for t, _ in a[4]] == [t for t, _ in b[4]]
    assert all(torch.equal(x[1], y[1]) for x, y in zip(a[4], b[4]))
    assert rel(a[0].float().cpu(), g["tower_out"]) < 1.2e-2 and rel(a[2][0].float().cpu(), g["mm_features"]) < 2.5e-2
    # the producer-side finalize (VL2_STAGE_ROW_TICKET: out_proj / fc2 / o / down write (mean, rstd) of their rows themselves, k_gemm.h
    # gemm_rows_ticket) is the same arithmetic as the launches: tower and prefill logits bit for bit, twice on the self-re-arming ticket block
    try:
        ops.set_stage_flags(ops.STAGE_ROW_TICKET)
        for _ in range(2):
            assert torch.equal(m.vision_tower(g["frames"].to(DEV)), a[0])
            assert torch.equal(dec.prefill(g["inputs_embeds"].to(DEV)), a[3])
        # VL2_STAGE_NO_MFMA16: gate/up on the family's 32 x 32 x 16 instruction instead of the default's 16 x 16 x 32 (k_gemm9.h; where the model's width has the
        # kernel built) -- other last bits than the default's, but the SAME bits from the C++ layer loop and from the per-operator loop, within the bf16 noise
        ops.set_stage_flags(ops.STAGE_NO_MFMA16)
        l16 = dec.prefill(g["inputs_embeds"].to(DEV)).clone()
        ops.STAGE_ABI = False
        assert torch.equal(dec.prefill(g["inputs_embeds"].to(DEV)), l16) and rel(l16, a[3]) < 5e-3
    finally:
        ops.STAGE_ABI = True
        ops.set_stage_flags(0)
    # the graph decoder.generate uses replays the same entry point: tokens equal to eager generate
    ids = g["input_ids"][None].to(DEV)
    kw = dict(attention_mask=torch.ones_like(ids), images=[(g["frames"].to(DEV), "video")], do_sample=False, max_new_tokens=6)
    assert m.generate(ids, use_graph=True, **kw)[0].tolist() == m.generate(ids, use_graph=False, **kw)[0].tolist()


def test_stage_calls_validate_their_arguments():
    import ctypes

    from videollama2_amd import _lib, ops
    lib = _lib.load()
    bad = _lib.VitDesc()                                   # size field 0: another ABI
    assert lib.vl2_vit_workspace_bytes(ctypes.byref(bad), 4) == -1
    x = torch.zeros(8, dtype=torch.uint8, device=DEV)
    with pytest.raises(_lib.Vl2HipError, match="descriptor"):
        _lib.call("vl2_vit_forward", ctypes.byref(bad), ops._p(x), 0, None, 1, ops._p(x), ops._p(x), 8, None)
    sbad = _lib.StcDesc()
    assert lib.vl2_stc_workspace_bytes(ctypes.byref(sbad), 4, 4, 8) == -1
    with pytest.raises(_lib.Vl2HipError, match="descriptor"):
        _lib.call("vl2_stc_forward", ctypes.byref(sbad), ops._p(x), 1, 2, ops._p(x), 1, 1, 1, ops._p(x), ops._p(x), 8, None)
    lbad = _lib.LlmDesc()
    assert lib.vl2_llm_workspace_bytes(ctypes.byref(lbad), 4) == -1
