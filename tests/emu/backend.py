"""TEST INFRASTRUCTURE ONLY: route videollama2_amd's C-ABI calls to the host emulator build of the SAME kernel
sources (tests/emu/_emu_kernels.so) so the real host code can be exercised on CPU tensors.  Never used by the
product; the product's `_lib.load()` only ever opens videollama2_amd/libvl2hip.so."""
import contextlib
import ctypes

from tests.emu.build_emu import build


@contextlib.contextmanager
def emulated_backend():
    from videollama2_amd import _lib, ops
    lib = ctypes.CDLL(build())
    lib.vl2_version.restype = ctypes.c_int32
    lib.vl2_last_error_string.restype = ctypes.c_char_p
    lib.vl2_workspace_bytes.restype = ctypes.c_int64
    lib.vl2_vit_workspace_bytes.restype = ctypes.c_int64
    lib.vl2_vit_workspace_bytes.argtypes = [ctypes.POINTER(_lib.VitDesc), ctypes.c_int32]
    lib.vl2_stc_workspace_bytes.restype = ctypes.c_int64
    lib.vl2_stc_workspace_bytes.argtypes = [ctypes.POINTER(_lib.StcDesc), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.vl2_llm_workspace_bytes.restype = ctypes.c_int64
    lib.vl2_llm_workspace_bytes.argtypes = [ctypes.POINTER(_lib.LlmDesc), ctypes.c_int32]
    for name, args in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int32
        fn.argtypes = args
    _lib.bind_lab(lib)                 # the emulator exports the lab entry points too (vl2_attn_decode_fused, vl2_decode_tail)
    saved = (_lib._lib, ops._chk, ops._stream)
    _lib._lib = lib

    def chk(t, dtype, name):
        if t is not None and t.dtype != dtype:
            raise TypeError(f"{name} must be {dtype}, got {t.dtype}")

    ops._chk = chk
    ops._stream = lambda: None
    try:
        yield lib
    finally:
        _lib._lib, ops._chk, ops._stream = saved
