"""ctypes binding of libctc_amd.so (C ABI declared in include/ctc_amd.h).

The product path has NO fallback: if the shared library is missing or does not load, importing the
symbols raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc,
--offload-arch=gfx950); the .so lives in-tree next to this file.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTC_AMD_LIB", os.path.join(_HERE, "libctc_amd.so"))  # override: kernel experiments only

ABI_VERSION = 5
CLASSIC, SIMPLIFIED = 0, 1
WRT_LOGITS, WRT_LOGPROBS = 0, 1
WS_LOSS_GRAD, WS_ALPHA_BETA, WS_HESSIAN, WS_HVP, WS_LOSS_GRAD_LOGITS = 0, 1, 2, 3, 4
OK, EINVAL, EWORKSPACE, EHIP, ELABEL = 0, -1, -2, -3, -4
F32, BF16, F16 = 0, 1, 2

_c_int, _c_void_p, _c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

_COMMON = [_c_int, _c_int,            # kind, wrt
           _c_void_p, _c_void_p, _c_int,  # logits, labels, label_stride
           _c_void_p, _c_void_p, _c_int,  # label_length, logit_length, blank_index
           _c_int, _c_int, _c_int, _c_int]  # B, T, V, U

# every symbol include/ctc_amd.h declares, with its argument types
SIGNATURES = {
    "ctc_amd_abi_version": (_c_int, []),
    "ctc_amd_last_error": (ctypes.c_char_p, []),
    "ctc_amd_pipeline_name": (ctypes.c_char_p, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ctc_amd_debug_override": (_c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "ctc_amd_debug_flags_offset": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_size_t)]),
    "ctc_amd_debug_hvp_flags_offset": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_size_t)]),
    "ctc_amd_reduce_loss": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "ctc_amd_probe_copy": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_probe_spin": (_c_int, [_c_int, _c_int, ctypes.c_float, _c_void_p]),
    "ctc_amd_check_labels": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "ctc_amd_workspace_bytes": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_size_t)]),
    "ctc_amd_loss_grad": (_c_int, _COMMON + [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_loss_grad_ex": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,  # kind, wrt, logits, dtype, strides
                                      _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,                 # labels .. blank_index
                                      _c_int, _c_int, _c_int, _c_int,                                   # B, T, V, U
                                      _c_void_p, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,    # loss, grad, dtype, strides
                                      _c_void_p, _c_void_p, _c_size_t, _c_void_p]),                     # d_loss, ws, bytes, stream
    "ctc_amd_loss_grad_packed": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, _c_void_p, ctypes.c_int64,        # kind, wrt, logits, dtype, row_offsets, row_stride
                                          _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,                  # labels .. blank_index
                                          _c_int, _c_int, _c_int, _c_int,                                    # B, T, V, U
                                          _c_void_p, _c_void_p, _c_int, ctypes.c_int64,                      # loss, grad, dtype, grad row stride
                                          _c_void_p, _c_void_p, _c_size_t, _c_void_p]),                      # d_loss, ws, bytes, stream
    "ctc_amd_loss_grad_sum": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,
                                       _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                                       _c_int, _c_int, _c_int, _c_int,
                                       _c_void_p, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),  # d_loss, sum2, zero_next, ws, bytes, stream
    "ctc_amd_loss_forward": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,
                                      _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                                      _c_int, _c_int, _c_int, _c_int,
                                      _c_void_p, _c_void_p, _c_size_t, _c_void_p]),                     # loss, ws, bytes, stream
    "ctc_amd_grad_resume": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,
                                     _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                                     _c_int, _c_int, _c_int, _c_int,
                                     _c_void_p, _c_void_p, _c_int, ctypes.c_int64, ctypes.c_int64,
                                     _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_alpha_beta": (_c_int, _COMMON + [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_hessian": (_c_int, _COMMON + [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_log_posterior": (_c_int, _COMMON + [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "ctc_amd_hvp": (_c_int, _COMMON + [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
}

_lib = None
override_generation = 0  # bumped by debug_override: cached pipeline names / workspace sizes of ops.py are keyed on it


class CtcAmdError(RuntimeError):
    """An entry point of libctc_amd.so returned a negative code."""


def load() -> ctypes.CDLL:
    """Loads libctc_amd.so (once) and binds every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root (needs hipcc). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.ctc_amd_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libctc_amd.so ABI version {got} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == OK:
        return
    msg = load().ctc_amd_last_error().decode("utf-8", "replace")
    if rc in (EINVAL, ELABEL):
        raise ValueError(f"{what}: {msg}")
    raise CtcAmdError(f"{what} failed with code {rc}: {msg}")


def pipeline_name(kind: int, wrt: int, B: int, T: int, V: int, U: int, want_grad: bool = True) -> str:
    return load().ctc_amd_pipeline_name(kind, wrt, B, T, V, U, int(want_grad)).decode()


def debug_override(key: str, value: str = "") -> None:
    """Diagnostic (parity tests, benchmarks): force a lower kernel tier ("pipeline": "v1" | "fused5") or the
    general Hessian kernel ("hessian": "slab"); the empty string restores the default.  Process-wide."""
    global override_generation
    check(load().ctc_amd_debug_override(key.encode(), value.encode()), "ctc_amd_debug_override")
    override_generation += 1


def flags_offset(kind: int, B: int, T: int, V: int, U: int) -> int:
    """Diagnostic: where the linear-domain kernel's per-utterance flag words sit in a logits call's workspace."""
    out = _c_size_t(0)
    check(load().ctc_amd_debug_flags_offset(kind, B, T, V, U, ctypes.byref(out)), "ctc_amd_debug_flags_offset")
    return int(out.value)


def hvp_flags_offset(kind: int, B: int, T: int, V: int, U: int) -> int:
    """Diagnostic: where the fused Hessian-vector kernel's per-utterance flag words sit in a WS_HVP workspace."""
    out = _c_size_t(0)
    check(load().ctc_amd_debug_hvp_flags_offset(kind, B, T, V, U, ctypes.byref(out)), "ctc_amd_debug_hvp_flags_offset")
    return int(out.value)


def workspace_bytes(what: int, kind: int, B: int, T: int, V: int, U: int) -> int:
    out = _c_size_t(0)
    check(load().ctc_amd_workspace_bytes(what, kind, B, T, V, U, ctypes.byref(out)), "ctc_amd_workspace_bytes")
    return int(out.value)
