"""Producer formats of round 3: float16 logits / gradient and packed (ragged) batches -- both read
and written by the three-kernel pipeline.  The reference takes contiguous float32 [B,T,V] only (base_loss.py:59,131), so the
checks are: the loss is the float32 loss of the same (rounded) values, the gradient is the float32 gradient rounded once to the
output type, and a packed batch gives, utterance by utterance, what the padded batch gives (to the last bits: the calls may run
different emission kernels) -- plus the float64 C oracle."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _case(B, T, U, V, seed):
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    tl = rng.integers(max(T // 2, 1), T + 1, B).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    tl[0], ll[0] = T, U
    return logits, labels, ll, tl


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,U,V,time_major", [(5, 40, 12, 64, False), (3, 33, 70, 300, True), (2, 20, 5, 2048, False), (4, 17, 9, 31, True)])
def test_float16_logits_and_gradient(kind, B, T, U, V, time_major):
    from tf_seq2seq_losses_amd import ops, _lib
    logits, labels, ll, tl = _case(B, T, U, V, seed=B + T)
    xh = _t(logits).to(torch.float16)
    if time_major:  # [T,B,V] storage, passed as a [B,T,V] view
        xh = xh.transpose(0, 1).contiguous().transpose(0, 1)
    x32 = xh.to(torch.float32).contiguous()
    k = ops.KINDS[kind]
    p16 = ops.Prepared(_t(labels), xh, _t(ll), _t(tl), 0, U=U, keep_format=True)
    assert p16.native and p16.x.dtype == torch.float16
    loss16, grad16 = ops.loss_grad(k, _lib.WRT_LOGITS, p16, True)
    assert grad16.dtype == torch.float16 and grad16.stride() == xh.stride()
    _lib.debug_override("pipeline", "v1")
    try:
        p32 = ops.Prepared(_t(labels), x32, _t(ll), _t(tl), 0, U=U)
        loss32, grad32 = ops.loss_grad(k, _lib.WRT_LOGITS, p32, True)
    finally:
        _lib.debug_override("pipeline", "")
    # the same float32 arithmetic on the same values, rounded once on the way out (not bit for bit: aligned float32 rows take the
    # four-frames-per-wavefront emission kernel / the 1024-column gradient kernel, whose sums associate differently)
    assert torch.allclose(loss16, loss32, rtol=2e-6, atol=0)
    assert torch.allclose(grad16.float(), grad32, rtol=1.0 / 1024, atol=1e-6)
    rl, rg = C.loss_grad(kind, labels, x32.cpu().numpy(), ll, tl, 0)
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(loss16.cpu().numpy()), fin)
    assert np.abs(loss16.cpu().numpy()[fin] - rl[fin]).max() < 1e-4 * max(1.0, np.abs(rl[fin]).max())
    assert np.abs(grad16.float().cpu().numpy() - rg).max() < 1e-3  # half precision of values <= 1


def test_float16_through_the_public_functions():
    import tf_seq2seq_losses_amd as ctc
    B, T, U, V = 4, 30, 8, 50
    logits, labels, ll, tl = _case(B, T, U, V, seed=3)
    x = _t(logits).to(torch.float16).requires_grad_(True)
    loss = ctc.classic_ctc_loss(_t(labels), x, _t(ll), _t(tl), 0)
    (g,) = torch.autograd.grad(loss.sum(), x)
    assert g.dtype == torch.float16 and g.shape == x.shape
    rl, rg = C.loss_grad("classic", labels, x.detach().float().cpu().numpy(), ll, tl, 0)
    assert np.abs(loss.detach().cpu().numpy() - rl).max() < 1e-4 * np.abs(rl).max()
    assert np.abs(g.float().cpu().numpy() - rg).max() < 1e-3


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,U,V,dtype", [(6, 50, 14, 40, torch.float32), (4, 37, 30, 2052, torch.float32), (5, 29, 7, 257, torch.float32),
                                            (3, 64, 20, 96, torch.bfloat16), (3, 21, 4, 33, torch.float16)])
def test_packed_batches(kind, B, T, U, V, dtype):
    """Utterance b owns rows off[b] .. off[b] + logit_length[b] - 1 of a [total, row_stride] tensor (here with a gap between
    utterances and a row stride wider than V): loss and gradient equal the padded call's, rows nobody owns are not touched."""
    from tf_seq2seq_losses_amd import ops, _lib
    logits, labels, ll, tl = _case(B, T, U, V, seed=11 * B + V)
    tl[1] = 0  # an utterance without frames
    xpad = _t(logits).to(dtype)
    gap = 3
    off = np.zeros(B, np.int64)
    total = 0
    for b in range(B):
        off[b] = total + gap
        total = int(off[b]) + int(tl[b])
    stride = V + 4
    store = torch.full((total + gap, stride), 7.0, dtype=dtype, device=_dev())
    packed = store[:, :V]
    for b in range(B):
        packed[off[b]:off[b] + tl[b]] = xpad[b, :tl[b]]
    k = ops.KINDS[kind]
    d_loss = _t(np.random.default_rng(1).standard_normal(B).astype(np.float32))
    loss, grad = ops.loss_grad_packed(k, _lib.WRT_LOGITS, _t(labels), packed, _t(off), _t(ll), _t(tl), 0, T, U=U, d_loss=d_loss)
    _lib.debug_override("pipeline", "v1")
    try:
        pp = ops.Prepared(_t(labels), xpad, _t(ll), _t(tl), 0, U=U, keep_format=True)
        loss_p, grad_p = ops.loss_grad(k, _lib.WRT_LOGITS, pp, True, d_loss=d_loss)
    finally:
        _lib.debug_override("pipeline", "")
    # (not bit for bit: aligned float32 rows of the padded call take the four-frames-per-wavefront emission kernel)
    assert torch.allclose(loss, loss_p, rtol=2e-6, atol=0)
    rt = {torch.float32: 2e-5, torch.bfloat16: 1.0 / 128, torch.float16: 1.0 / 1024}[dtype]
    owned = torch.zeros(total + gap, dtype=torch.bool, device=_dev())
    for b in range(B):
        assert torch.allclose(grad[off[b]:off[b] + tl[b]].float(), grad_p[b, :tl[b]].float(), rtol=rt, atol=1e-6)
        owned[off[b]:off[b] + tl[b]] = True
    assert not grad[~owned].any()  # (zero-initialised by the wrapper, never written by the kernels)
    rl, rg = C.loss_grad(kind, labels, xpad.float().cpu().numpy(), ll, tl, 0)
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(loss.cpu().numpy()), fin)
    tol = TOL if dtype == torch.float32 else 1e-2 if dtype == torch.bfloat16 else 1e-3
    for b in range(B):
        assert np.abs(grad[off[b]:off[b] + tl[b]].float().cpu().numpy() - rg[b, :tl[b]] * float(d_loss[b])).max(initial=0.0) < tol * max(1.0, abs(float(d_loss[b])))


def test_packed_argument_errors():
    from tf_seq2seq_losses_amd import _lib
    lib = _lib.load()
    one = torch.zeros(64, device=_dev())
    ptr = one.data_ptr()
    rc = lib.ctc_amd_loss_grad_packed(0, 0, ptr, _lib.F32, None, 8, ptr, 1, ptr, ptr, 0, 1, 1, 8, 1, ptr, ptr, _lib.F32, 8, None, ptr, 1 << 20, None)
    assert rc == _lib.EINVAL and b"row_offsets" in lib.ctc_amd_last_error()
    rc = lib.ctc_amd_loss_grad_packed(0, 0, ptr, _lib.F32, ptr, 4, ptr, 1, ptr, ptr, 0, 1, 1, 8, 1, ptr, ptr, _lib.F32, 8, None, ptr, 1 << 20, None)
    assert rc == _lib.EINVAL and b"row strides" in lib.ctc_amd_last_error()
    rc = lib.ctc_amd_loss_grad_packed(0, 0, ptr, 3, ptr, 8, ptr, 1, ptr, ptr, 0, 1, 1, 8, 1, ptr, ptr, _lib.F32, 8, None, ptr, 1 << 20, None)
    assert rc == _lib.EINVAL and b"dtype" in lib.ctc_amd_last_error()
