"""ctypes binding of libvl2hip.so (include/vl2hip.h).  The product path has NO fallback: if the HIP library is
missing or a call fails, this raises -- nothing here (or anywhere in videollama2_amd/) imports oracle/."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvl2hip.so")

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class GemmDesc(ctypes.Structure):
    """include/vl2hip.h `vl2_gemm_desc` (field for field)."""
    _fields_ = [("size", ctypes.c_uint32), ("M", _i32), ("N", _i32), ("K", _i32),
                ("A", _vp), ("lda", _i32), ("W", _vp), ("ldw", _i32), ("C", _vp), ("ldc", _i32),
                ("bias", _vp), ("res", _vp), ("ldres", _i32), ("act", _i32), ("flags", _i32),
                ("a_idx", _vp), ("seg_k", _i32),
                ("out_grp", _i32), ("out_grp_pad", _i32), ("out_row_off", _i32), ("res_row_mod", _i32), ("res_row_off", _i32),
                ("stats_out", _vp), ("stats_in", _vp), ("norm", _i32), ("norm_eps", _f32), ("w_colsum", _vp), ("row_norm", _vp),
                ("ws", _vp), ("ws_bytes", _i64), ("variant", _i32), ("col_scale", _vp), ("tile_ctr", _vp),
                ("row_norm_out", _vp), ("row_ticket", _vp), ("norm_out", _i32), ("norm_out_eps", _f32)]


class VitLayer(ctypes.Structure):
    """include/vl2hip.h `vl2_vit_layer`."""
    _fields_ = [(n, _vp) for n in ("wqkv", "bqkv", "sqkv", "wo", "bo", "w1", "b1", "s1", "w2", "b2")]


class VitDesc(ctypes.Structure):
    """include/vl2hip.h `vl2_vit_desc`."""
    _fields_ = [("size", ctypes.c_uint32), ("family", _i32), ("image", _i32), ("patch", _i32), ("D", _i32), ("I", _i32), ("heads", _i32),
                ("head_dim", _i32), ("n_layers", _i32), ("kp", _i32), ("act", _i32), ("eps", _f32), ("attn_scale", _f32),
                ("patch_w", _vp), ("patch_b", _vp), ("pos", _vp), ("cls_pos", _vp), ("pre_w", _vp), ("pre_b", _vp),
                ("layers", ctypes.POINTER(VitLayer)), ("flags", ctypes.c_uint32)]


class StcBlock(ctypes.Structure):
    """include/vl2hip.h `vl2_stc_block`."""
    _fields_ = [(n, _vp) for n in ("conv1_w", "bn1_w", "bn1_b", "dw_w", "bn2_w", "bn2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                   "conv3_w", "bn3_w", "bn3_b", "ds_w", "dsbn_w", "dsbn_b")] + [("rd", _i32)]


class StcDesc(ctypes.Structure):
    """include/vl2hip.h `vl2_stc_desc`."""
    _fields_ = [("size", ctypes.c_uint32), ("cin", _i32), ("C", _i32), ("s1", StcBlock * 4), ("s2", StcBlock * 4),
                ("samp_w", _vp), ("samp_b", _vp), ("ro0_w", _vp), ("ro0_b", _vp), ("ro2_w", _vp), ("ro2_b", _vp), ("flags", ctypes.c_uint32)]


class LlmLayer(ctypes.Structure):
    """include/vl2hip.h `vl2_llm_layer`."""
    _fields_ = [(n, _vp) for n in ("wqkv", "bqkv", "wo", "wgu", "wd", "kcache", "vcache")]


class LlmLayerW8(ctypes.Structure):
    """include/vl2hip.h `vl2_llm_layer_w8` (fp8 copies of a layer's packed projections + their row scales)."""
    _fields_ = [(n, _vp) for n in ("wqkv", "sqkv", "wo", "so", "wgu", "sgu", "wd", "sd")]


class LlmDesc(ctypes.Structure):
    """include/vl2hip.h `vl2_llm_desc`."""
    _fields_ = [("size", ctypes.c_uint32), ("D", _i32), ("I", _i32), ("heads", _i32), ("kv_heads", _i32), ("n_layers", _i32), ("vocab", _i32),
                ("smax", _i32), ("eps", _f32), ("layers", ctypes.POINTER(LlmLayer)), ("embed", _vp), ("norm_w", _vp), ("ones", _vp),
                ("lm_head", _vp), ("cos_t", _vp), ("sin_t", _vp), ("flags", ctypes.c_uint32),
                ("layers_w8", ctypes.POINTER(LlmLayerW8)), ("lm_head_w8", _vp), ("lm_head_scale", _vp)]


# name -> argtypes (all return int32 except the two below)
SIGNATURES = {
    "vl2_gemm": [ctypes.POINTER(GemmDesc), _vp],
    "vl2_row_stats": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vl2_fill_zero": [_vp, _i64, _vp],
    "vl2_row_norm_finalize": [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "vl2_vit_forward": [ctypes.POINTER(VitDesc), _vp, _i32, _vp, _i32, _vp, _vp, _i64, _vp],
    "vl2_stc_forward": [ctypes.POINTER(StcDesc), _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp],
    "vl2_llm_prefill": [ctypes.POINTER(LlmDesc), _vp, _i32, _vp, _vp, _i64, _vp],
    "vl2_llm_decode_step": [ctypes.POINTER(LlmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "vl2_pack_fold_norm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "vl2_pack_gate_up": [_vp, _vp, _vp, _i32, _i32, _vp],
    "vl2_pack_permute": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vl2_pack_pad_rows": [_vp, _vp, _i64, _i64, _i64, _vp],
    "vl2_pack_cvt_f32": [_vp, _vp, _i64, _vp],
    "vl2_pack_quant_fp8": [_vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "vl2_quant_act_fp8": [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _vp],
    "vl2_gemv_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp],
    "vl2_layernorm": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "vl2_rmsnorm": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "vl2_patchify": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "vl2_patchify_u8": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "vl2_fill_cls": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vl2_attn_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _i32, _i32,
                     _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp],
    "vl2_dwconv3x3_ln_silu": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "vl2_chan_mean": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vl2_small_linear": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vl2_se_scale": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vl2_dwconv3x3_ln_silu_mean": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp],
    "vl2_se_excite_scale": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vl2_rope_kv": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "vl2_gemv_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp],
    "vl2_gemm_skinny_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "vl2_gemv_batched_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "vl2_attn_decode": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _vp],
    "vl2_attn_decode_batched": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _i32, _f32, _vp],
    "vl2_argmax": [_vp, _i32, _vp, _vp, _i32, _vp, _vp],
    "vl2_sample_token": [_vp, _i32, _f32, _i32, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "vl2_embed_rows": [_vp, _vp, _vp, _i32, _i32, _i32, _vp],
}
# entry points of the LAB build only (libvl2hip_lab.so = the same sources with -DVL2_LAB, scripts/build_lab_lib.sh: the experiments that were
# measured and lost stay buildable and testable without riding in the product library); also exported by the CPU emulator (tests/emu)
LAB_SIGNATURES = {
    "vl2_decode_tail": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp],
    "vl2_attn_decode_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _f32, _vp, _vp],
}
EXPORTS = ["vl2_version", "vl2_elem_name", "vl2_last_error_string", "vl2_workspace_bytes", "vl2_vit_workspace_bytes", "vl2_stc_workspace_bytes", "vl2_llm_workspace_bytes", "vl2_dwconv_mean_workspace_bytes"] + list(SIGNATURES)

_lib = None
# ---- element type of the build in use: "bf16" (libvl2hip.so, the default: BASELINE.json configs[1]) or "fp16" (libvl2hip_f16.so = the same
# sources with -DVL2_ELEM_F16; the reference's own dtype, videollama2/__init__.py:60 `.half().cuda()`).  Process-wide host-side setting:
# choose it BEFORE building models (their packed weights and buffers are allocated in it).
_ELEM = "bf16"
_LIBS = {}
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_HERE, "libvl2hip_f16.so")}
LAB_LIB_PATH = os.path.join(_HERE, "libvl2hip_lab.so")       # bf16 elements
_LAB = False


def lab_built():
    return os.path.exists(LAB_LIB_PATH)


def lab():
    return _LAB


def set_lab(on):
    """Route every kernel call that follows through the LAB build (bf16 only; scripts, A/B runs and the lab tests).  The product never
    calls this."""
    global _LAB, _lib
    on = bool(on)
    if on == _LAB:
        return
    if on and _ELEM != "bf16":
        raise ValueError("the lab library is built on bf16 elements")
    if on and not lab_built():
        raise Vl2HipError(f"{LAB_LIB_PATH} not found: build it with scripts/build_lab_lib.sh")
    if _lib is not None:
        _LIBS["lab" if _LAB else _ELEM] = _lib
    _LAB = on
    _lib = _LIBS.get("lab" if on else _ELEM)


def elem():
    return _ELEM


def elem_dtype():
    import torch
    return torch.float16 if _ELEM == "fp16" else torch.bfloat16


def check_elem(built_in, what):
    """A module's packed weights, caches and stage descriptors are in the element type that was current when it was BUILT; the kernels
    cannot tell bf16 bits from fp16 bits, so a module called after `set_elem` switched the process to the other build refuses loudly."""
    if built_in != _ELEM:
        raise RuntimeError(f"{what} was built with element type {built_in!r} but the library in use is {_ELEM!r} "
                           f"(_lib.set_elem): rebuild the module, or switch back with _lib.set_elem({built_in!r})")


def set_elem(name):
    """Select the 16-bit element type ("bf16" | "fp16") of every kernel call that follows."""
    global _ELEM, _lib
    if name not in LIB_PATHS:
        raise ValueError(f"unknown element type {name!r} (bf16 | fp16)")
    if name == _ELEM:
        return
    if _LAB:
        raise ValueError("switch the lab library off (_lib.set_lab(False)) before changing the element type")
    if _lib is not None:
        _LIBS[_ELEM] = _lib
    _ELEM = name
    _lib = _LIBS.get(name)


class Vl2HipError(RuntimeError):
    pass


def load():
    """Load libvl2hip.so (once).  Raises if it has not been built: there is deliberately no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = LAB_LIB_PATH if _LAB else (LIB_PATHS[_ELEM] if _ELEM != "bf16" else LIB_PATH)
    if not os.path.exists(path):
        raise Vl2HipError(f"{path} not found: build it with `python -m videollama2_amd.csrc.build` "
                          "(hipcc --offload-arch=gfx950); the HIP path has no fallback")
    lib = ctypes.CDLL(path)
    lib.vl2_version.restype = _i32
    lib.vl2_version.argtypes = []
    lib.vl2_last_error_string.restype = ctypes.c_char_p
    lib.vl2_last_error_string.argtypes = []
    lib.vl2_workspace_bytes.restype = _i64
    lib.vl2_workspace_bytes.argtypes = []
    lib.vl2_vit_workspace_bytes.restype = _i64
    lib.vl2_vit_workspace_bytes.argtypes = [ctypes.POINTER(VitDesc), _i32]
    lib.vl2_stc_workspace_bytes.restype = _i64
    lib.vl2_stc_workspace_bytes.argtypes = [ctypes.POINTER(StcDesc), _i32, _i32, _i32]
    lib.vl2_dwconv_mean_workspace_bytes.restype = _i64
    lib.vl2_dwconv_mean_workspace_bytes.argtypes = [_i32, _i32]
    lib.vl2_llm_workspace_bytes.restype = _i64
    lib.vl2_llm_workspace_bytes.argtypes = [ctypes.POINTER(LlmDesc), _i32]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _i32
        fn.argtypes = args
    bind_lab(lib)
    if lib.vl2_version() != 7:
        raise Vl2HipError("libvl2hip.so ABI version mismatch")
    lib.vl2_elem_name.restype = ctypes.c_char_p
    lib.vl2_elem_name.argtypes = []
    if lib.vl2_elem_name().decode() != _ELEM:
        raise Vl2HipError(f"{path} computes in {lib.vl2_elem_name().decode()}, not {_ELEM}: rebuild it (python -m videollama2_amd.csrc.build)")
    _lib = lib
    _LIBS["lab" if _LAB else _ELEM] = lib
    return lib


def bind_lab(lib):
    """ctypes signatures of the lab entry points a library exports (the lab build, the emulator); returns their names."""
    have = []
    for name, args in LAB_SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = _i32
        fn.argtypes = args
        have.append(name)
    return have


def call(name, *args):
    """Call an entry point; non-zero return codes become Vl2HipError with the library's message."""
    lib = load()
    if name in LAB_SIGNATURES and not hasattr(lib, name):
        raise Vl2HipError(f"{name} is a lab entry point: built into libvl2hip_lab.so only (scripts/build_lab_lib.sh, then _lib.set_lab(True))")
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.vl2_last_error_string().decode(errors="replace")
        raise Vl2HipError(f"{name} failed (rc={rc}): {msg}")
