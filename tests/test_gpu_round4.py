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
    (fewer than 64 spare frames) and not inside its three bounds; a stand-alone loss-only call honours them always."""
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


def test_recorded_case_of_the_eight_positions_per_lane_roles():
    """r03's soak (seed 43) recorded one excess over 1e-4 and left it as found: tests/golden/soak_case_redone_nl8.npz -- classic,
    291 frames for 284 labels (label bound 300: eight label positions per lane), V = 64, sharp logits; the linear-domain kernel
    flagged it (D7|D4|D1) and the log-domain roles' gradient was 1.05e-4 off (the alignment is binding and sharp, eight states share
    a lane's exponent, so the linear domain cannot hold it; the log-domain roles' float32 log-sum-exp chain accumulated the rest).
    Since r04 the log-domain roles keep their lattice state in float64 and the case holds north_star's 1e-4."""
    import os
    from tf_seq2seq_losses_amd import ops, _lib
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "soak_case_redone_nl8.npz"), allow_pickle=True)
    x, labels, ll, tl, U = d["x"], d["labels"], d["ll"], d["tl"], int(d["U"])
    rl, rg = C.loss_grad("classic", labels, x, ll, tl, 0)
    p = ops.Prepared(_t(labels), _t(x), _t(ll), _t(tl), 0, U=U)
    assert ops.pipeline_of(0, _lib.WRT_LOGITS, p) == "fused6"
    loss, grad = ops.loss_grad(0, _lib.WRT_LOGITS, p, True)
    assert abs(float(loss[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))
    assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4
    loss2, ws2 = ops.loss_forward(0, _lib.WRT_LOGITS, p)
    grad2 = ops.grad_resume(0, _lib.WRT_LOGITS, p, ws2)
    assert abs(float(loss2[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))
    assert np.abs(grad2.cpu().numpy() - rg).max() < 1e-4


def test_posterior_products_do_not_overflow_unnoticed():
    """tests/golden/r04_case_product_overflow.npz (sigma 5, V = 3, 32 labels in 51 frames): with live lanes lifted only to their
    neighbour's exponent - 2^80 (an r04 experiment) a chain's mantissas reached 2^122, the posterior products of phase 2 overflowed
    between the frames the mass check samples, and the gradient came back 3.0 off with NO flag.  The gap is the adoption gap again
    (mantissas below 2^55); the case must give the oracle's gradient."""
    import os
    from tf_seq2seq_losses_amd import ops, _lib
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r04_case_product_overflow.npz"), allow_pickle=True)
    x, labels, ll, tl, U = d["x"], d["labels"], d["ll"], d["tl"], int(d["U"])
    rl, rg = C.loss_grad("classic", labels, x, ll, tl, 0)
    for bound in (U, 128):  # one and two label positions per lane
        lab = np.zeros((1, bound), np.int32); lab[:, :labels.shape[1]] = labels
        p = ops.Prepared(_t(lab), _t(x), _t(ll), _t(tl), 0, U=bound)
        loss, grad = ops.loss_grad(0, _lib.WRT_LOGITS, p, True)
        assert abs(float(loss[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))
        assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_second_order_entry_points_take_offset_views(kind):
    """A batch-sliced view whose byte offset is not a multiple of 16 (T * V * 4 = 60 here): ctc_amd_hvp / ctc_amd_hessian state 16-byte
    alignment as a requirement; the Python front end copies such a view once instead of raising."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(9)
    B, T, V, U = 4, 5, 3, 2
    big = _t(rng.standard_normal((B + 1, T, V)).astype(np.float32))
    vbig = _t(rng.standard_normal((B + 1, T, V)).astype(np.float32))
    x, v = big[1:], vbig[1:]
    assert x.data_ptr() % 16 != 0 and v.data_ptr() % 16 != 0
    labels = _t(rng.integers(1, V, (B, U)).astype(np.int32))
    ll, tl = _t(np.full(B, U, np.int32)), _t(np.full(B, T, np.int32))
    k = ops.KINDS[kind]
    p_view = ops.Prepared(labels, x, ll, tl, 0, U=U)
    p_copy = ops.Prepared(labels, x.clone(), ll, tl, 0, U=U)
    out_view = ops.hvp(k, _lib.WRT_LOGITS, p_view, v)
    out_copy = ops.hvp(k, _lib.WRT_LOGITS, p_copy, v.clone())
    assert torch.equal(out_view[-1] if isinstance(out_view, tuple) else out_view, out_copy[-1] if isinstance(out_copy, tuple) else out_copy)
    h_view = ops.hessian(k, _lib.WRT_LOGITS, p_view)
    h_copy = ops.hessian(k, _lib.WRT_LOGITS, p_copy)
    for a, b in zip(h_view, h_copy):
        if a is not None:
            assert torch.equal(a, b)


def test_label_bound_from_a_host_label_length_costs_no_sync():
    """A label tensor wider than 128 with label_length on the HOST: the bound is taken from the caller's tensor (no device round trip,
    no cache); with a device tensor the cache is keyed on the caller's own object, also when it needs a dtype conversion."""
    from tf_seq2seq_losses_amd import ops
    rng = np.random.default_rng(1)
    B, T, V, W = 3, 40, 16, 200
    x = _t(rng.standard_normal((B, T, V)).astype(np.float32))
    labels = _t(rng.integers(1, V, (B, W)).astype(np.int32))
    ll_host = torch.tensor([5, 9, 7], dtype=torch.int64)           # host, and not int32
    p = ops.Prepared(labels, x, ll_host, torch.full((B,), T, dtype=torch.int32), 0)
    assert p.U == 9
    ll_dev = ll_host.to(_dev())                                      # device int64: converted inside, cached on THIS object
    ops._MAXLEN_CACHE.clear()
    p1 = ops.Prepared(labels, x, ll_dev, torch.full((B,), T, dtype=torch.int32), 0)
    assert p1.U == 9 and id(ll_dev) in ops._MAXLEN_CACHE
    n = len(ops._MAXLEN_CACHE)
    ops.Prepared(labels, x, ll_dev, torch.full((B,), T, dtype=torch.int32), 0)
    assert len(ops._MAXLEN_CACHE) == n


def test_forward_half_keeps_every_sign_with_many_label_positions_per_lane():
    """tests/golden/r04_case_forward_loss_nl8.npz (r04 soak, seed 101): 7 labels in 158 frames, V = 300, logits N(0, 3^2), label bound
    300 -> eight label positions per lane.  Not a binding alignment, so the first r04 version of ctc_amd_loss_forward let the linear
    sweeps' loss stand: 1343.19 for 1341.21 (the resume call's mass check then redid the gradient).  The relaxed rule is for one and
    two positions per lane only."""
    import os
    from tf_seq2seq_losses_amd import ops, _lib
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r04_case_forward_loss_nl8.npz"), allow_pickle=True)
    x, labels, ll, tl, U = d["x"], d["labels"], d["ll"], d["tl"], int(d["U"])
    rl, rg = C.loss_grad("classic", labels, x, ll, tl, 0)
    p = ops.Prepared(_t(labels), _t(x), _t(ll), _t(tl), 0, U=U)
    loss, ws = ops.loss_forward(0, _lib.WRT_LOGITS, p)
    grad = ops.grad_resume(0, _lib.WRT_LOGITS, p, ws)
    assert abs(float(loss[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))
    assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4


@pytest.mark.gpu
def test_loss_only_calls_with_long_dwell_on_mild_logits_take_the_log_domain():
    """2-3 labels in 512 frames under a label bound of 128 (two label positions per lane), N(0, 2^2) logits over 29 tokens: the
    soft signs D3/D4/D5/D7 miss 22 % of these utterances, and r03 / early r04 builds returned 2 of 256 losses more than 1e-4 off
    (tests/tools/flag_stats_short_labels.py).  D10 (> 40 frames per label position, loss-only calls) sends all of them to the
    log-domain roles."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(5)
    B, T, U, V = 256, 512, 128, 29
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    ll = rng.integers(2, 4, B).astype(np.int32)
    tl = np.full(B, T, np.int32)
    logits = (rng.standard_normal((B, T, V)) * 2.0).astype(np.float32)
    rl, _ = C.loss_grad("classic", labels, logits, ll, tl, 0)
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
    ws = ops._workspace(_lib.WS_LOSS_GRAD_LOGITS, 0, p)
    loss, _ = ops.loss_grad(0, _lib.WRT_LOGITS, p, False, workspace=ws)
    fl = ops.fused_flags(ws, 0, p).cpu().numpy()
    assert (fl & 2048).all(), np.unique(fl)
    assert (np.abs(loss.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4
    # the forward half of a pair: same rule (dwell > 12), same accuracy
    loss2, ws2 = ops.loss_forward(0, _lib.WRT_LOGITS, p)
    assert (np.abs(loss2.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4


def test_forward_half_trusts_the_linear_sweeps_only_inside_its_three_bounds():
    """ctc_amd_loss_forward lets the linear sweeps' loss stand only where the mass check of calls with a gradient finds nothing to
    redo: >= 64 spare frames, <= 12 frames per label position, P decaying by <= 10 bits per frame (11.75 on the simplified lattice)
    (and <= 2 label positions per lane).  Outside any of the three it keeps every soft sign, like a stand-alone call.  Cases:
    N(0, 5^2) logits over 256 tokens (decay 15 bits per frame), 2 labels in 199 frames (dwell 66: tests/golden/r04_case_forward_loss_dwell.npz, r04 soak seed 106 -- the state that
    carries the posterior sits 2^-105 below a lane-mate, forward loss 1235.84 for 1234.70), and N(0, 3^2) inside the bounds."""
    import os
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(21)
    B, T, U, V = 48, 400, 100, 256
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    ll, tl = np.full(B, U, np.int32), np.full(B, T, np.int32)
    for sigma, trusted in ((5.0, False), (3.0, True)):
        logits = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
        p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
        loss, ws = ops.loss_forward(0, _lib.WRT_LOGITS, p)
        fl = ops.fused_flags(ws, 0, p).cpu().numpy()
        grad = ops.grad_resume(0, _lib.WRT_LOGITS, p, ws)
        rl, rg = C.loss_grad("classic", labels, logits, ll, tl, 0)
        assert (np.abs(loss.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4
        assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4
        if trusted:
            assert not fl.any(), np.unique(fl)
        else:
            assert (fl & 128).all(), np.unique(fl)
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r04_case_forward_loss_dwell.npz"), allow_pickle=True)
    x, lab, ll1, tl1, U1 = d["x"], d["labels"], d["ll"], d["tl"], int(d["U"])
    rl, rg = C.loss_grad("classic", lab, x, ll1, tl1, 0)
    p = ops.Prepared(_t(lab), _t(x), _t(ll1), _t(tl1), 0, U=U1)
    loss, ws = ops.loss_forward(0, _lib.WRT_LOGITS, p)
    grad = ops.grad_resume(0, _lib.WRT_LOGITS, p, ws)
    assert abs(float(loss[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))
    assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4
    # a stand-alone loss-only call on the same utterance: more than 40 frames per label position is a sign of its own (D10 = 2048)
    ws = ops._workspace(_lib.WS_LOSS_GRAD_LOGITS, 0, p)
    loss1, _ = ops.loss_grad(0, _lib.WRT_LOGITS, p, False, workspace=ws)
    assert int(ops.fused_flags(ws, 0, p)[0]) & 2048
    assert abs(float(loss1[0]) - rl[0]) < 1e-4 * max(1.0, abs(rl[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_log_domain_tiers_hold_1e_4_on_sharp_and_long_inputs(kind):
    """The three-kernel pipeline (vocabularies beyond 1024) and the log-domain roles of the fused tier on the inputs where their float32
    log-sum-exp chain used to exceed north_star's tolerance: N(0, 4^2) logits at T = 1000 (r03 / early r04: 1.4e-4 / 1.9e-4 for the
    three kernels, 1.2-1.3e-4 for the roles) and T = 3000 (1.4e-4).  Their lattice state is float64 since r04 (ctc_common.h lse2,
    ctc_v1_device.h Scan, ctc_fused_common.h Side<..., double>): measured 2.5e-5 ... 3.5e-5."""
    from tf_seq2seq_losses_amd import ops, _lib
    k = ops.KINDS[kind]
    for (B, T, U, V, sigma, force) in ((4, 1000, 128, 2048, 4.0, ""), (2, 3000, 128, 2048, 1.0, ""), (6, 1000, 128, 256, 4.0, "fused5")):
        rng = np.random.default_rng(3)
        x = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
        labels = rng.integers(1, V, (B, U)).astype(np.int32)
        ll, tl = np.full(B, U, np.int32), np.full(B, T, np.int32)
        rl, rg = C.loss_grad(kind, labels, x, ll, tl, 0)
        _lib.debug_override("pipeline", force)
        try:
            p = ops.Prepared(_t(labels), _t(x), _t(ll), _t(tl), 0, U=U)
            assert ops.pipeline_of(k, _lib.WRT_LOGITS, p) == (force or "v1")
            loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True)
        finally:
            _lib.debug_override("pipeline", "")
        assert (np.abs(loss.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))).max() < 1e-4
        assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4, (B, T, V, sigma, force, np.abs(grad.cpu().numpy() - rg).max())
