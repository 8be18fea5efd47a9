"""Producer formats: bfloat16 logits/gradient and time-major [T,B,V] activations
through ctc_amd_loss_grad_ex, without a conversion pass.

The reference reads a contiguous float32 [B,T,V] tensor only (base_loss.py:59,131).  Parity here:
  * the loss of a bfloat16 / strided call is BIT-IDENTICAL to the plain float32 call on the same values (the kernels
    widen on load and do the same float32 arithmetic);
  * the gradient is the plain call's gradient rounded to bfloat16 (<= 2^-8 relative) or, for strided float32, identical;
  * the plain call itself is compared with the oracle elsewhere (test_gpu_parity.py, test_gpu_large.py).
"""
import numpy as np
import pytest
import torch

from oracle import ctc_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _call(kind, labels, x, ll, tl):
    import tf_seq2seq_losses_amd as ctc
    fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simplified_ctc_loss
    x = x.detach().requires_grad_(True)
    loss = fn(labels, x, ll, tl, 0)
    fin = torch.isfinite(loss)
    (g,) = torch.autograd.grad(loss[fin].sum(), x)
    return loss.detach(), g


def _inputs(B, T, V, U, seed, ragged=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, V)).astype(np.float32) * 2
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    if ragged:
        tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32)
        ll = rng.integers(0, U + 1, B).astype(np.int32)
    else:
        tl, ll = np.full(B, T, np.int32), np.full(B, U, np.int32)
    return _t(x), _t(labels), _t(ll), _t(tl)


SHAPES = [(6, 40, 256, 20), (3, 150, 256, 128), (5, 33, 12, 9), (4, 64, 32, 70), (2, 50, 7, 5)]


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,V,U", SHAPES)
def test_time_major_view_is_zero_copy_and_identical(kind, B, T, V, U):
    """x_tm is [T,B,V]; x_tm.transpose(0,1) is a [B,T,V] view with strides (V, B*V, 1): same numbers as the plain call,
    and the gradient comes back with the same strides."""
    x, labels, ll, tl = _inputs(B, T, V, U, 1)
    x_tm = x.transpose(0, 1).contiguous()          # time-major storage
    view = x_tm.transpose(0, 1)                    # [B,T,V] view of it
    assert not view.is_contiguous()
    l0, g0 = _call(kind, labels, x, ll, tl)
    l1, g1 = _call(kind, labels, view, ll, tl)
    assert torch.equal(l0, l1)
    assert g1.stride() == view.stride()
    assert torch.equal(g0, g1)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,V,U", SHAPES)
@pytest.mark.parametrize("time_major", [False, True])
def test_bfloat16_logits_and_gradient(kind, B, T, V, U, time_major):
    x, labels, ll, tl = _inputs(B, T, V, U, 2)
    xb = x.to(torch.bfloat16)
    if time_major:
        xb = xb.transpose(0, 1).contiguous().transpose(0, 1)
    l0, g0 = _call(kind, labels, xb.float().contiguous(), ll, tl)   # plain float32 call on the same values
    l1, g1 = _call(kind, labels, xb, ll, tl)
    assert g1.dtype == torch.bfloat16 and g1.stride() == xb.stride()
    from tf_seq2seq_losses_amd import _lib
    same_kernel = V % 4 == 0 or _lib.pipeline_name(0, 0, B, T, V, U, True) == "v1"
    if same_kernel:  # both calls run the same kernel: same float32 arithmetic after widening, one rounding at the end
        assert torch.equal(l0, l1)
        assert torch.equal(g1, g0.to(torch.bfloat16))
    else:            # float32 on the fused kernel, bfloat16 (rows not 8-byte aligned) on the three-kernel pipeline
        assert torch.allclose(l0, l1, rtol=2e-6, atol=0)
        assert ((g1.float() - g0).abs() <= 2.0 ** -8 * g0.abs() + 1e-6).all()


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_bfloat16_against_oracle(kind):
    """End to end against the NumPy oracle on the bfloat16-rounded logits (loss 1e-4, gradient bfloat16 resolution)."""
    x, labels, ll, tl = _inputs(4, 30, 256, 12, 3)
    xb = x.to(torch.bfloat16)
    l1, g1 = _call(kind, labels, xb, ll, tl)
    xr = xb.float().cpu().numpy()
    ref = O.ctc_loss(kind, labels.cpu().numpy(), xr, ll.cpu().numpy(), tl.cpu().numpy(), 0)
    fin = np.isfinite(ref.loss)
    assert np.array_equal(np.isfinite(l1.cpu().numpy()), fin)
    assert np.abs(l1.cpu().numpy()[fin] - ref.loss[fin]).max() < 1e-4 * max(1.0, np.abs(ref.loss[fin]).max())
    gref = O.logits_gradient(ref, xr)
    assert np.abs(g1.float().cpu().numpy() - gref).max() < 2.0 ** -8


def test_north_star_shape_bf16_time_major():
    """B=16 of the north-star shape in the layout an acoustic model hands over: [T,B,V] bfloat16."""
    B, T, V, U = 16, 1000, 256, 128
    x, labels, ll, tl = _inputs(B, T, V, U, 4, ragged=False)
    xb = x.to(torch.bfloat16).transpose(0, 1).contiguous().transpose(0, 1)
    l0, g0 = _call("classic", labels, xb.float().contiguous(), ll, tl)
    l1, g1 = _call("classic", labels, xb, ll, tl)
    assert torch.equal(l0, l1)
    assert torch.equal(g1, g0.to(torch.bfloat16))
    from tf_seq2seq_losses_amd import _lib
    assert _lib.pipeline_name(0, 0, B, T, V, U, True) == "fused6"


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_second_order_with_bfloat16_logits(kind):
    """Double backward on bfloat16 logits: the HVP runs in float32 on the widened logits and is rounded once."""
    import tf_seq2seq_losses_amd as ctc
    x, labels, ll, tl = _inputs(3, 20, 8, 6, 5, ragged=False)
    fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simplified_ctc_loss
    v = torch.randn(x.shape, device=_dev(), generator=torch.Generator(device=_dev()).manual_seed(0))

    def second(xin):
        xin = xin.detach().requires_grad_(True)
        loss = fn(labels, xin, ll, tl, 0)
        (g,) = torch.autograd.grad(loss.sum(), xin, create_graph=True)
        (h,) = torch.autograd.grad((g.float() * v).sum(), xin)
        return h

    xb = x.to(torch.bfloat16)
    h0 = second(xb.float())
    h1 = second(xb)
    assert h1.dtype == torch.bfloat16
    assert (h1.float() - h0).abs().max().item() <= 2.0 ** -7 * max(1.0, h0.abs().max().item())


def test_ex_entry_validation():
    from tf_seq2seq_losses_amd import _lib
    lib = _lib.load()
    x = torch.zeros((2, 4, 8), device=_dev())
    lab = torch.ones((2, 2), dtype=torch.int32, device=_dev())
    n = torch.full((2,), 2, dtype=torch.int32, device=_dev())
    t = torch.full((2,), 4, dtype=torch.int32, device=_dev())
    loss = torch.empty(2, device=_dev())
    ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, 2, 4, 8, 2), dtype=torch.uint8, device=_dev())

    def call(dtype, sb, st):
        return lib.ctc_amd_loss_grad_ex(0, 0, x.data_ptr(), dtype, sb, st, lab.data_ptr(), 2, n.data_ptr(), t.data_ptr(), 0,
                                        2, 4, 8, 2, loss.data_ptr(), None, 0, 32, 8, None, ws.data_ptr(), ws.numel(), None)
    assert call(0, 32, 8) == _lib.OK
    assert call(7, 32, 8) == _lib.EINVAL          # unknown element type
    assert call(0, 32, 4) == _lib.EINVAL          # rows would overlap
