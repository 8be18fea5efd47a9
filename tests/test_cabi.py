"""CPU-side checks of the drop-in boundary: libctc_amd.so loads, exports every symbol that include/ctc_amd.h
declares, reports the header's ABI version, validates arguments without touching a GPU, and the product
package never imports the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from tf_seq2seq_losses_amd import _lib
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ctc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctc_amd_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from tf_seq2seq_losses_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ctc_amd.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_matches_header(lib):
    text = open(os.path.join(ROOT, "include", "ctc_amd.h")).read()
    ver = int(re.search(r"#define CTC_AMD_ABI_VERSION (\d+)", text).group(1))
    assert lib.ctc_amd_abi_version() == ver


def test_workspace_bytes_and_argument_errors(lib):
    from tf_seq2seq_losses_amd import _lib
    n = _lib.workspace_bytes(_lib.WS_LOSS_GRAD, _lib.CLASSIC, 256, 1000, 256, 128)
    # emissions [B,T,UP+4] + alpha, beta [B,T+1,2*UP+4] + logp
    assert n >= 256 * 1000 * 132 * 4 + 2 * 256 * 1001 * 264 * 4
    assert _lib.workspace_bytes(_lib.WS_HESSIAN, _lib.SIMPLIFIED, 2, 5, 3, 4) > _lib.workspace_bytes(_lib.WS_LOSS_GRAD, _lib.SIMPLIFIED, 2, 5, 3, 4)
    # the workspace of a logits call is the selected pipeline's own: checkpoint rows only on the fused tiers (<= 64 MB at the
    # north star), the full lattice rows where the three-kernel pipeline runs (long labels, BPE-sized vocabularies)
    small = _lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, _lib.CLASSIC, 256, 1000, 256, 128)
    assert small <= 64 << 20 < n
    assert _lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, 4, 100, 256, 1000) == _lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, 4, 100, 256, 1000)
    assert _lib.flags_offset(0, 256, 1000, 256, 128) < small
    with pytest.raises(ValueError):
        _lib.flags_offset(0, 4, 100, 256, 1000)  # three-kernel pipeline: no flags
    with pytest.raises(ValueError):
        _lib.workspace_bytes(7, 0, 1, 1, 1, 1)
    with pytest.raises(ValueError):
        _lib.workspace_bytes(0, 0, 1, 1, 1, 100000)  # U beyond the supported maximum
    # bad kind is rejected before anything is launched (no GPU needed for the validation path)
    rc = lib.ctc_amd_loss_grad(5, 0, None, None, 0, None, None, 0, 1, 1, 3, 1, None, None, None, None, 0, None)
    assert rc == _lib.EINVAL and b"kind" in lib.ctc_amd_last_error()
    rc = lib.ctc_amd_loss_grad(0, 0, None, None, 0, None, None, 9, 1, 1, 3, 1, None, None, None, None, 0, None)
    assert rc == _lib.EINVAL and b"blank" in lib.ctc_amd_last_error()
    # B == 0 is not an error (tests/test_classic_ctc_loss.py:309-330)
    assert lib.ctc_amd_loss_grad(0, 0, None, None, 0, None, None, 0, 0, 4, 3, 2, None, None, None, None, 0, None) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tf_seq2seq_losses_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_cpu_tensors_fail_loudly():
    import torch
    import tf_seq2seq_losses_amd as ctc
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ctc.classic_ctc_loss(torch.zeros((1, 2), dtype=torch.int32), torch.zeros((1, 3, 4)),
                             torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))


def test_limits_are_reported_not_crashed_into(lib):
    """include/ctc_amd.h "Limits": U <= 1024, V <= 16384 (loss / gradient), V <= 16380 (Hessian, HVP): exactly at the limit
    the argument checks pass (here they then stop at the missing workspace), one past it they return CTC_AMD_EINVAL."""
    from tf_seq2seq_losses_amd import _lib
    text = open(os.path.join(ROOT, "include", "ctc_amd.h")).read()
    lim = {k: int(v) for k, v in re.findall(r"#define (CTC_AMD_MAX_[A-Z_]+) (\d+)", text)}
    assert lim == {"CTC_AMD_MAX_U": 1024, "CTC_AMD_MAX_V": 16384, "CTC_AMD_MAX_V_HESSIAN": 16380}
    one = ctypes.c_void_p(16)  # a non-null, never dereferenced pointer: validation happens before any launch

    def loss_grad(U, V, grad=one):
        return lib.ctc_amd_loss_grad(0, 0, one, one, U, one, one, 0, 1, 1, V, U, one, grad, None, None, 0, None)

    def hessian(V):
        return lib.ctc_amd_hessian(0, 0, one, one, 1, one, one, 0, 1, 1, V, 1, one, None, one, None, 0, None)

    def hvp(V):
        return lib.ctc_amd_hvp(0, 0, one, one, 1, one, one, 0, 1, 1, V, 1, one, one, None, one, None, 0, None)

    assert loss_grad(1024, 8) == _lib.EWORKSPACE and loss_grad(1025, 8) == _lib.EINVAL
    assert b"U=1025" in lib.ctc_amd_last_error()
    assert loss_grad(4, 16384) == _lib.EWORKSPACE and loss_grad(4, 16385) == _lib.EINVAL
    assert loss_grad(4, 20000, grad=None) == _lib.EWORKSPACE        # the loss alone has no vocabulary limit
    assert hessian(16380) == _lib.EWORKSPACE and hessian(16381) == _lib.EINVAL
    assert hvp(16380) == _lib.EWORKSPACE and hvp(16381) == _lib.EINVAL
    with pytest.raises(ValueError):
        _lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, 1, 1, 8, 1025)


def test_debug_override_validation(lib):
    from tf_seq2seq_losses_amd import _lib
    for key, val in (("pipeline", "v1"), ("pipeline", "fused5"), ("pipeline", ""), ("hessian", "slab"), ("hessian", "")):
        _lib.debug_override(key, val)
    assert _lib.pipeline_name(0, 0, 256, 1000, 256, 128) == "fused6"
    # BPE-sized vocabularies: the three kernels (the experimental one-launch tier of csrc/ctc_wide.hip exists in diagnostic builds only)
    assert _lib.pipeline_name(0, 0, 32, 1000, 4096, 128) == "v1"
    with pytest.raises(ValueError):
        _lib.debug_override("pipeline", "wide")
    _lib.debug_override("pipeline", "fused5")
    try:
        assert _lib.pipeline_name(0, 0, 256, 1000, 256, 128) == "fused5"
        assert _lib.pipeline_name(0, 0, 256, 1000, 2048, 128) == "v1"
    finally:
        _lib.debug_override("pipeline", "")
    for key, val in (("pipeline", "fused"), ("pipeline", "f"), ("hessian", "x"), ("nope", "")):
        with pytest.raises(ValueError):
            _lib.debug_override(key, val)
    # the library never reads the environment
    for f in os.listdir(os.path.join(ROOT, "tf_seq2seq_losses_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "tf_seq2seq_losses_amd", "csrc", f)).read(), f
