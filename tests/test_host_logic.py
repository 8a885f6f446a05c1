"""CPU tests of the host-side mirror of the reference interface (no kernels involved) and of the C-ABI library's
export table (loads libvl2hip.so; no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from videollama2_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "vl2hip.h")).read()
    # the lab entry points (libvl2hip_lab.so only) are declared inside #ifdef VL2_EXPERIMENTAL blocks: the product exports exactly the rest
    product_part = re.sub(r"#ifdef VL2_EXPERIMENTAL.*?#endif /\* VL2_EXPERIMENTAL[^\n]*", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vl2_[a-z0-9_]+)\s*\(", product_part))
    lab_declared = set(re.findall(r"\b(vl2_[a-z0-9_]+)\s*\(", header)) - declared
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lab_declared == set(_lib.LAB_SIGNATURES), lab_declared ^ set(_lib.LAB_SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    for name in lab_declared:
        assert not hasattr(lib, name), f"{name}: a lab entry point exported by the product library"
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l and l.split()[-1].startswith("vl2_")}
    assert exported == declared, exported ^ declared                  # `nm -D` = the documented export set
    assert _lib.load().vl2_version() == 7
    assert _lib.load().vl2_elem_name() == b"bf16"
    f16 = ctypes.CDLL(_lib.LIB_PATHS["fp16"])               # the fp16 build: same export table, other element type
    for name in declared:
        assert hasattr(f16, name), name
    f16.vl2_elem_name.restype = ctypes.c_char_p
    assert f16.vl2_elem_name() == b"fp16"


def test_abi_argument_validation_without_gpu():
    from videollama2_amd import _lib
    def desc(**kw):
        d = _lib.GemmDesc()
        d.size = ctypes.sizeof(_lib.GemmDesc)
        for k, v in kw.items():
            setattr(d, k, v)
        return ctypes.byref(d)

    with pytest.raises(_lib.Vl2HipError, match="null pointer"):
        _lib.call("vl2_gemm", desc(M=1, N=128, K=64, lda=64, ldw=64, ldc=128), None)
    with pytest.raises(_lib.Vl2HipError, match="N%128"):
        _lib.call("vl2_gemm", desc(A=16, W=16, C=16, M=4, N=100, K=64, lda=64, ldw=64, ldc=104), None)
    with pytest.raises(_lib.Vl2HipError, match="another ABI"):            # a caller built against another struct layout
        d = _lib.GemmDesc()
        d.size = 8
        _lib.call("vl2_gemm", ctypes.byref(d), None)
    with pytest.raises(_lib.Vl2HipError, match="stats_in"):               # a fused norm without the producer's statistics
        _lib.call("vl2_gemm", desc(A=16, W=16, C=16, M=4, N=128, K=64, lda=64, ldw=64, ldc=128, norm=1), None)
    with pytest.raises(_lib.Vl2HipError, match="w_colsum"):
        _lib.call("vl2_gemm", desc(A=16, W=16, C=16, M=4, N=128, K=64, lda=64, ldw=64, ldc=128, norm=2, stats_in=16), None)
    with pytest.raises(_lib.Vl2HipError, match="workspace"):
        _lib.call("vl2_gemm", desc(A=16, W=16, C=16, M=4, N=128, K=64, lda=64, ldw=64, ldc=128, ws=16, ws_bytes=64), None)
    with pytest.raises(_lib.Vl2HipError, match="workspace"):
        _lib.call("vl2_gemm_skinny_bf16", 16, 16, 16, None, None, 4, 128, 64, 64, 64, 128, 0, 0, None, 0, None)
    with pytest.raises(_lib.Vl2HipError, match="head_dim"):
        _lib.call("vl2_attn_fwd", 16, 16, 16, 16, 0, 80, 80, 0, 80, 80, 0, 80, 80, 0, 80, 80, 1, 1, 4, 4, 1, 1.0, 0, 0, 80, 0, None)
    with pytest.raises(_lib.Vl2HipError, match="C%64"):
        _lib.call("vl2_row_stats", 16, 16, 4, 100, 104, None)
    # fp8 entry points: shapes the kernels cannot take are refused before anything is launched
    with pytest.raises(_lib.Vl2HipError, match="K%16"):
        _lib.call("vl2_pack_quant_fp8", 16, 4, 40, 40, 16, 16, None)
    with pytest.raises(_lib.Vl2HipError, match="N even"):
        _lib.call("vl2_gemv_fp8", 16, 16, 16, None, None, None, 16, 7, 64, 64, 1e-5, 0, None)
    with pytest.raises(_lib.Vl2HipError, match="SWIGLU"):
        _lib.call("vl2_gemv_fp8", 16, 16, 16, None, None, None, 16, 96, 64, 64, 1e-5, 1, None)
    d = _lib.LlmDesc()
    d.size = ctypes.sizeof(_lib.LlmDesc) - 8                                # a caller built against the v3 descriptor (no fp8 fields)
    with pytest.raises(_lib.Vl2HipError, match="another ABI"):
        _lib.call("vl2_llm_decode_step", ctypes.byref(d), 16, 16, 16, 16, 16, 16, 64, None)


def test_library_keeps_no_mutable_process_state():
    """'no global mutable state except the error string (thread-local)'.  The header offers no setter, and the
    translation unit defines no writable namespace-scope object besides thread-locals and the per-kernel LDS-attribute bits."""
    header = open(os.path.join(ROOT, "include", "vl2hip.h")).read()
    assert "vl2_set_" not in header
    src = open(os.path.join(ROOT, "videollama2_amd", "csrc", "vl2_abi.hip")).read()
    globals_ = re.findall(r"^static\s+(?!thread_local|inline|const|constexpr|int32_t\s+\w+\(|bool\s+\w+\(|void\s+\w+\(|int\s+\w+\(|GemmArgs\s+\w+\()([^;(]+);", src, re.M)
    assert not globals_, globals_


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "videollama2_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f"{f} imports the oracle"
                assert "tests.emu" not in txt and "hip_emu" not in txt, f"{f} references the emulator"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from videollama2_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Vl2HipError, match="no fallback"):
        _lib.load()


def test_frame_sample_and_process_video_match_reference_goldens(golden_small):
    from videollama2_amd.mm_utils import frame_sample, process_video
    from videollama2_amd.tower import default_image_processor
    g = golden_small
    for (d, n), ids in g["frame_sample"].items():
        assert np.array_equal(frame_sample(d, "uniform", num_frames=n), ids.numpy())
    assert frame_sample(90, "fps", fps=30).tolist() == [15, 45, 75]
    proc = default_image_processor(g["cfg"]["vision"]["image_size"])
    T = g["cfg"]["num_frames"]
    out = process_video(g["frames_u8"].numpy(), proc, aspect_ratio=None, num_frames=T)
    assert out.dtype == torch.float32 and torch.allclose(out, g["frames"], atol=1e-6)
    odd = g["odd_u8"].numpy()
    out = process_video([f for f in odd], proc, aspect_ratio=None, num_frames=T)         # bicubic resize + crop + pad frame
    assert torch.allclose(out, g["odd_frames"], atol=1e-6)
    out = process_video(odd, proc, aspect_ratio="pad", num_frames=T)
    assert torch.allclose(out, g["odd_frames_pad"], atol=1e-6)
    with pytest.raises(ValueError, match="Unsupported video path type"):
        process_video(123, proc)


def test_tokenizer_multimodal_token_and_stopping_criteria():
    from videollama2_amd.mm_utils import KeywordsStoppingCriteria, tokenizer_multimodal_token

    class Tok:
        bos_token_id = 1

        def __call__(self, text, add_special_tokens=True):
            ids = [10 + (ord(c) % 50) for c in text]
            return type("E", (), {"input_ids": ([1] + ids) if add_special_tokens else ids})()

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["".join(chr(97 + int(t) % 26) for t in row) for row in ids]

    tok = Tok()
    ids = tokenizer_multimodal_token("ab<video>\ncd", tok, "<video>", return_tensors="pt")
    assert ids.tolist() == tok("ab", False).input_ids + [-201] + tok("\ncd", False).input_ids
    assert tokenizer_multimodal_token("xyz", tok, "") == tok("xyz", False).input_ids
    crit = KeywordsStoppingCriteria(["q"], tok, torch.zeros(1, 3, dtype=torch.long))
    kw = crit.keyword_ids[0]
    assert crit(torch.cat([torch.tensor([5, 6, 7, 8]), kw])[None], None)
    assert not crit(torch.tensor([[5, 6, 7, 8, 9]]), None) or "q" in tok.batch_decode(torch.tensor([[5, 6, 7, 8, 9]]))[0]


def test_conv3d_gather_table_equals_conv3d():
    """The gathered-GEMM formulation of Conv3d(k=2,s=2,p=1) (projector.py:164-174) on the host, in fp64."""
    from videollama2_amd.connector import conv3d_k2s2p1_index
    for T, H in ((16, 24), (5, 7), (2, 4)):
        C, Co = 6, 5
        x = torch.randn(T * H * H, C, dtype=torch.float64)
        w = torch.randn(Co, C, 2, 2, 2, dtype=torch.float64)
        idx, (To, Ho, Wo) = conv3d_k2s2p1_index(T, H, H, "cpu")
        wp = w.permute(0, 2, 3, 4, 1).reshape(Co, 8 * C)
        rows = torch.cat([torch.where((idx[s] >= 0)[:, None], x[idx[s].clamp(min=0).long()], torch.zeros(1, C, dtype=torch.float64))
                          for s in range(8)], 1)
        ref = F.conv3d(x.view(1, T, H, H, C).permute(0, 4, 1, 2, 3), w, stride=2, padding=1)[0].permute(1, 2, 3, 0)
        assert (To, Ho, Wo) == tuple(ref.shape[:3]) == (T // 2 + 1, H // 2 + 1, H // 2 + 1)
        assert torch.allclose(rows @ wp.T, ref.reshape(-1, Co), atol=1e-10)


def test_gate_up_packing_roundtrip():
    from videollama2_amd.weights import pack_gate_up
    g, u = torch.arange(64 * 4.).view(64, 4), -torch.arange(64 * 4.).view(64, 4)
    p = pack_gate_up(g, u)
    assert torch.equal(p[:32], g[:32]) and torch.equal(p[32:64], u[:32]) and torch.equal(p[64:96], g[32:])


def test_weight_key_normalisation_accepts_4_40_names():
    from videollama2_amd.weights import normalise_keys
    sd = {"model.vision_tower.vision_tower.vision_model.embeddings.class_embedding": 1, "lm_head.weight": 2}
    assert "model.vision_tower.vision_tower.embeddings.class_embedding" in normalise_keys(sd)


def test_frame_sharder_split():
    from videollama2_amd.dist import FrameSharder
    assert FrameSharder.split(32, 8) == [(i * 4, 4) for i in range(8)]
    assert FrameSharder.split(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert FrameSharder.split(1, 2) == [(0, 1), (1, 0)]
    assert sum(c for _, c in FrameSharder.split(16, 3)) == 16


def test_config_checks():
    from videollama2_amd.config import check_supported, videollama2_7b
    check_supported(videollama2_7b(16))
    bad = videollama2_7b(16)
    bad["llm"]["head_dim"] = 96
    with pytest.raises(ValueError, match="head_dim"):
        check_supported(bad)


def test_checkpoint_config_uses_public_tower_values(tmp_path):
    """api.config_from_checkpoint: hub tower names map to the public hyper-parameters (no hub access on the box), the model type
    selects the decoder family, unknown towers / model types raise like the reference's factories."""
    import json
    from videollama2_amd import api
    from videollama2_amd.config import check_supported, videollama2_1_7b_16f, videollama2_7b
    base = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_frames=16, mm_projector_type="stc_connector",
                model_type="videollama2_mistral", mm_vision_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2)
    json.dump(base, open(tmp_path / "config.json", "w"))
    cfg, _ = api.config_from_checkpoint(str(tmp_path))
    want = videollama2_7b(16)
    assert cfg["llm"] == {**want["llm"], "family": "mistral"} and cfg["vision"] == {**want["vision"], "family": "clip"}
    check_supported(cfg)
    q = dict(base, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
             vocab_size=152064, rms_norm_eps=1e-6, model_type="videollama2_qwen2", mm_projector_type="stc_connector_v35",
             mm_vision_tower="google/siglip-so400m-patch14-384")
    json.dump(q, open(tmp_path / "config.json", "w"))
    cfg, _ = api.config_from_checkpoint(str(tmp_path))
    want = videollama2_1_7b_16f(16)
    assert cfg["llm"] == want["llm"] and cfg["vision"] == want["vision"] and cfg["projector"] == "stc_connector_v35"
    check_supported(cfg)
    json.dump(dict(base, mm_vision_tower="someone/some-other-tower"), open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="Unknown vision tower"):
        api.config_from_checkpoint(str(tmp_path))
    json.dump(dict(base, model_type="videollama2_mixtral"), open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="not built"):
        api.config_from_checkpoint(str(tmp_path))


def test_lazy_tower_key_layouts_and_never_loaded_guard(tmp_path):
    """(1) a local tower directory whose config.json nests the vision hyper-parameters under `vision_config`
    (the standard openai/clip-vit-large-patch14-336 layout) is read; (2) a `vision_model.`-prefixed (transformers 4.x era) checkpoint
    loads through HF `from_pretrained(low_cpu_mem_usage=True)` -- which matches keys by NAME and runs no load_state_dict hook --
    when the tower hosts that layout; (3) with the mismatching layout the parameters are materialised uninitialised and
    `check_loaded()` (called by `pack()`) refuses to run instead of encoding frames with garbage; (4) a plain `load_state_dict`
    accepts either layout."""
    import json
    import types
    from safetensors.torch import save_file
    from transformers import PretrainedConfig, PreTrainedModel
    from videollama2_amd.lazy import LazyHipVisionTower, default_key_layout, tower_config
    tdir = tmp_path / "clip-tiny"
    tdir.mkdir()
    vis = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14)
    json.dump({"model_type": "clip", "text_config": {}, "vision_config": vis}, open(tdir / "config.json", "w"))
    v = tower_config(str(tdir))
    assert v["family"] == "clip" and all(v[k] == vis[k] for k in vis)
    assert default_key_layout() in ("vision_model", "flat")
    args = types.SimpleNamespace(mm_vision_select_layer=-2)

    class Cfg(PretrainedConfig):
        model_type = "vl2_lazy_probe"

        def __init__(self, key_layout="vision_model", **kw):
            super().__init__(**kw)
            self.key_layout = key_layout

    class Host(PreTrainedModel):
        config_class = Cfg

        def __init__(self, config):
            super().__init__(config)
            self.vision_tower = LazyHipVisionTower(str(tdir), args, key_layout=config.key_layout)
            self.post_init()

        def _init_weights(self, module):
            pass

    ref = LazyHipVisionTower(str(tdir), args, key_layout="vision_model")
    with pytest.raises(RuntimeError, match="never loaded"):
        ref.check_loaded()                                                         # fresh torch.empty storage is not a tower
    g = torch.Generator().manual_seed(0)
    sd = {"vision_tower." + k: torch.randn(p.shape, generator=g).bfloat16() for k, p in ref.state_dict().items()}
    assert all(".vision_model." in k for k in sd)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file(sd, str(ck / "model.safetensors"))
    Cfg(key_layout="vision_model").save_pretrained(str(ck))
    m = Host.from_pretrained(str(ck), low_cpu_mem_usage=True)
    m.vision_tower.check_loaded()
    for k, p in m.vision_tower.state_dict().items():
        assert torch.equal(p.bfloat16(), sd["vision_tower." + k]), k
    Cfg(key_layout="flat").save_pretrained(str(ck))
    m2 = Host.from_pretrained(str(ck), low_cpu_mem_usage=True)                      # every tower key is reported missing
    with pytest.raises(RuntimeError, match="key_layout='vision_model'"):
        m2.vision_tower.check_loaded()
    flat = LazyHipVisionTower(str(tdir), args, key_layout="flat")
    flat.load_state_dict({k[len("vision_tower."):]: t for k, t in sd.items()}, strict=True)     # 4.x keys into the 5.x host
    flat.check_loaded()
    ref.load_state_dict({k: p for k, p in flat.state_dict().items()}, strict=True)                  # 5.x keys into the 4.x host
    ref.check_loaded()
    # tensors that arrive through load_state_dict are loaded by construction -- a synthetic tower of constants passes
    ones = LazyHipVisionTower(str(tdir), args, key_layout="vision_model")
    ones.load_state_dict({k: torch.ones_like(p) for k, p in ref.state_dict().items()}, strict=True)
    ones.check_loaded()


def test_loader_delegates_to_the_reference_class_only_on_evidence(tmp_path, monkeypatch):
    """ADVICE r03 (install.py): `HipCausalLMLoader.from_pretrained` hands a checkpoint back to the class it replaced only when a READABLE
    config lacks the multimodal keys (the LoRA / model_base branches of load_pretrained_model load plain LLM checkpoints); an unreadable
    config (hub id, offline) is delegated WITH a warning; a multimodal checkpoint never is -- bitsandbytes kwargs are refused there."""
    import json
    import warnings
    from videollama2_amd import install

    calls = []

    class Orig:
        @classmethod
        def from_pretrained(cls, path, *a, **kw):
            calls.append(str(path))
            return "reference-model"

    class Loader(install.HipCausalLMLoader):
        _orig = Orig

    plain = tmp_path / "plain-llm"
    plain.mkdir()
    json.dump({"model_type": "mistral", "hidden_size": 64}, open(plain / "config.json", "w"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # readable config without mm keys: silent delegation
        assert Loader.from_pretrained(str(plain)) == "reference-model"
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.warns(UserWarning, match="NOT run on the HIP kernels"):
        assert Loader.from_pretrained("some-org/some-hub-id-that-is-not-cached") == "reference-model"
    assert calls == [str(plain), "some-org/some-hub-id-that-is-not-cached"]
    mm = tmp_path / "mm-ckpt"
    mm.mkdir()
    json.dump({"model_type": "videollama2_mistral", "mm_vision_tower": "openai/clip-vit-large-patch14-336", "mm_projector_type": "stc_connector"},
              open(mm / "config.json", "w"))
    with pytest.raises(NotImplementedError, match="bitsandbytes"):
        Loader.from_pretrained(str(mm), load_in_4bit=True)         # a multimodal checkpoint stays on the HIP loader (and it refuses bnb kwargs)
    assert len(calls) == 2
