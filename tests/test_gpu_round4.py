"""Round-4 parity tests (need a real MI355X: `pytest -m gpu`):

* sharp logits N(0, 3^2) at the north-star shape stay in the linear domain -- one-call form and the public forward + backward form
  (loss-only call, then gradient resume) -- unflagged and within 1e-4 of the float64 C oracle (until r04 four
  of 256 such utterances overflowed a chain's mantissas and EVERY one of them took the log-domain roles in a loss-only call);
* a binding alignment with sharp logits is still handed to the log domain by a loss-only call, a non-binding one is not;
* the meeting-point products survive mantissas whose plain product underflows.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.as_tensor(np.asarray(a)).to(_dev())


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_sharp_logits_at_the_north_star_shape_stay_in_the_linear_domain(kind):
    import tf_seq2seq_losses_amd as ctc
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(0)
    B, T, U, V = 256, 1000, 128, 256
    logits = rng.standard_normal((B, T, V), dtype=np.float32) * np.float32(3.0)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    ll, tl = np.full(B, U, np.int32), np.full(B, T, np.int32)
    k = ops.KINDS[kind]
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, k, B, T, V, U), dtype=torch.uint8, device=_dev())
    loss1, grad1 = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
    fl1 = ops.fused_flags(ws, k, p).cpu().numpy()
    loss2, ws2 = ops.loss_forward(k, _lib.WRT_LOGITS, p)
    fl2a = ops.fused_flags(ws2, k, p).cpu().numpy().copy()
    grad2 = ops.grad_resume(k, _lib.WRT_LOGITS, p, ws2)
    fl2b = ops.fused_flags(ws2, k, p).cpu().numpy()
    assert not fl1.any(), ("one call", np.unique(fl1))
    assert not fl2a.any() and not fl2b.any(), ("two calls", np.unique(fl2a), np.unique(fl2b))
    n = 24  # (the oracle at this size: ~1 s per utterance and core)
    rl, rg = C.loss_grad(kind, labels[:n], logits[:n], ll[:n], tl[:n], 0)
    for loss, grad in ((loss1, grad1), (loss2, grad2)):
        assert (np.abs(loss[:n].cpu().numpy() - rl) / np.abs(rl)).max() < 1e-6
        assert np.abs(grad[:n].cpu().numpy() - rg).max() < 1e-4
    assert torch.equal(loss1, loss2)
    # the public functions (autograd forward + backward)
    fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simplified_ctc_loss
    x = _t(logits).requires_grad_(True)
    loss = fn(_t(labels), x, _t(ll), _t(tl), 0)
    (g,) = torch.autograd.grad(loss.mean(), x)
    assert np.abs(g[:n].cpu().numpy() * B - rg).max() < 1e-4


def test_loss_only_calls_send_binding_sharp_alignments_to_the_log_domain_and_no_others():
    """The first half of a forward / backward pair (ctc_amd_loss_forward) honours the soft signs D3 / D4 / D7 for binding alignments
    only (fewer than 64 spare frames); a stand-alone loss-only call honours them always."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(3)
    B, U, V = 64, 100, 64
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    rep = (labels[:, 1:] == labels[:, :-1]).sum(axis=1)
    ll = np.full(B, U, np.int32)
    for slack, expect_flagged in ((3, True), (200, False)):
        tl = (U + rep + slack).astype(np.int32)
        T = int(tl.max())
        logits = (rng.standard_normal((B, T, V)) * 3.0).astype(np.float32)
        p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
        loss, ws = ops.loss_forward(0, _lib.WRT_LOGITS, p)
        fl = ops.fused_flags(ws, 0, p).cpu().numpy()
        grad = ops.grad_resume(0, _lib.WRT_LOGITS, p, ws)
        rl, rg = C.loss_grad("classic", labels, logits, ll, tl, 0)
        assert (np.abs(loss.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4
        assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4
        if expect_flagged:
            assert (fl & 128).all(), np.unique(fl)       # sharp and binding: the log-domain roles
        else:
            assert not (fl & (28 | 128)).any(), np.unique(fl)  # sharp, not binding: the linear sweep's answer stands
        # a STAND-ALONE loss-only call (no resume will check the posterior mass) keeps every sign, binding or not
        ws3 = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, B, T, V, U), dtype=torch.uint8, device=_dev())
        loss3, _ = ops.loss_grad(0, _lib.WRT_LOGITS, p, False, workspace=ws3)
        assert (ops.fused_flags(ws3, 0, p).cpu().numpy() & 128).all()
        assert (np.abs(loss3.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4
