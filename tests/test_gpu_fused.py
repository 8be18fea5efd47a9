"""The fused loss+gradient kernels (csrc/ctc_fused6.hip = the default tier "", csrc/ctc_fused5.hip, csrc/ctc_fused.hip; logits input) against
the float64 C oracle and against the three-kernel pipeline (ctc_amd_debug_override("pipeline", "v1")), including the edge cases the reference tests:
ragged and zero lengths, infeasible samples, empty labels, repeated tokens, d_loss weighting."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run(kind, logits, labels, ll, tl, pipeline, d_loss=None):
    from tf_seq2seq_losses_amd import ops, _lib
    dev = torch.device("cuda:0")
    _lib.debug_override("pipeline", pipeline)
    try:
        p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(logits).to(dev), torch.from_numpy(ll).to(dev),
                         torch.from_numpy(tl).to(dev), 0)
        dl = None if d_loss is None else torch.from_numpy(d_loss).to(dev)
        loss, grad = ops.loss_grad(ops.KINDS[kind], _lib.WRT_LOGITS, p, True, d_loss=dl)
        torch.cuda.synchronize()
    finally:
        _lib.debug_override("pipeline", "")
    return loss.cpu().numpy(), grad.cpu().numpy()


def _check(kind, logits, labels, ll, tl, d_loss=None):
    l5, g5 = _run(kind, logits, labels, ll, tl, "fused5", d_loss)
    lf, gf = _run(kind, logits, labels, ll, tl, "", d_loss)  # default tier: fused6 (+ fused5 for what it flags)
    l1, g1 = _run(kind, logits, labels, ll, tl, "v1", d_loss)
    rl, rg = C.loss_grad(kind, labels, logits, ll, tl, 0)
    if d_loss is not None:
        rg = rg * d_loss[:, None, None]
    fin = np.isfinite(rl)
    for lo, gr, name in ((l5, g5, "fused5"), (lf, gf, "fused6"), (l1, g1, "v1")):
        assert np.array_equal(np.isfinite(lo), fin), name
        assert np.all(lo[~fin] == np.inf), name
        if fin.any():
            assert (np.abs(lo[fin] - rl[fin]) / np.maximum(1, np.abs(rl[fin]))).max() < TOL, name
        assert np.isfinite(gr).all(), name
        assert np.abs(gr - rg).max() < TOL, (name, np.abs(gr - rg).max())


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("V,U,T", [(256, 20, 70), (256, 100, 150), (512, 40, 97), (1024, 12, 40), (256, 200, 260)])
def test_fused_random_ragged(kind, V, U, T):
    rng = np.random.default_rng(V + U + T)
    B = 6
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    labels[1, : U // 2] = 7  # run of repeats
    tl = rng.integers(T // 2, T + 1, B).astype(np.int32)
    ll = rng.integers(U // 3, U + 1, B).astype(np.int32)
    tl[0], ll[0] = T, U
    _check(kind, logits, labels, ll, tl)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_fused_edge_lengths(kind):
    """logit_length 0/1/2/3, label_length 0, infeasible samples, logit_length > T (behaves as T), d_loss weights."""
    rng = np.random.default_rng(5)
    B, T, V, U = 10, 40, 256, 6
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    labels[6, :] = 9  # six repeats
    tl = np.array([0, 1, 2, 3, 40, 17, 8, 33, 100, 16], dtype=np.int32)
    ll = np.array([0, 1, 1, 6, 0, 3, 6, 6, 2, 5], dtype=np.int32)   # sample 3: too long; sample 6: classic needs 11 frames
    d_loss = rng.standard_normal(B).astype(np.float32)
    tl_ref = np.minimum(tl, T)
    rl, rg = C.loss_grad(kind, labels, logits, ll, tl_ref, 0)
    rg = rg * d_loss[:, None, None]
    fin = np.isfinite(rl)
    for pipeline in ("", "fused5", "v1"):
        lf, gf = _run(kind, logits, labels, ll, tl, pipeline, d_loss)
        assert np.array_equal(np.isfinite(lf), fin), pipeline
        assert (np.abs(lf[fin] - rl[fin]) / np.maximum(1, np.abs(rl[fin]))).max() < TOL, pipeline
        assert np.abs(gf - rg).max() < TOL, pipeline
        assert lf[0] == 0.0 and np.all(gf[0] == 0), pipeline  # T=0-like sample with empty label: loss 0


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_fused_neg_inf_logits(kind):
    rng = np.random.default_rng(9)
    B, T, V, U = 3, 30, 256, 5
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    logits[0, :, 100:] = -np.inf
    logits[1, 4, :] = -np.inf
    logits[1, 4, 0] = 0.0          # frame forced to blank
    labels = rng.integers(1, 100, (B, U)).astype(np.int32)
    ll = np.array([5, 3, 5], dtype=np.int32)
    tl = np.array([30, 30, 22], dtype=np.int32)
    _check(kind, logits, labels, ll, tl)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_loss_only_stops_at_the_meeting_point(kind):
    """grad == NULL (forward_fn alone, base_loss.py:140-155): fused5 runs phase 1 only; the loss must be the one the
    loss+gradient call returns, including +inf for infeasible samples and the ragged/empty cases."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(9)
    B, T, V, U = 9, 131, 256, 60
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    ll = np.array([60, 0, 13, 60, 1, 33, 60, 7, 20], np.int32)
    tl = np.array([131, 131, 40, 50, 0, 131, 12, 99, 131], np.int32)  # 50 < 60 and 12 < 60: infeasible
    dev = torch.device("cuda:0")
    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(logits).to(dev), torch.from_numpy(ll).to(dev),
                     torch.from_numpy(tl).to(dev), 0)
    assert _lib.pipeline_name(ops.KINDS[kind], _lib.WRT_LOGITS, B, T, V, U, False) == "fused6"
    l_only, g_none = ops.loss_grad(ops.KINDS[kind], _lib.WRT_LOGITS, p, False)
    l_both, _ = ops.loss_grad(ops.KINDS[kind], _lib.WRT_LOGITS, p, True)
    assert g_none is None
    assert torch.equal(l_only, l_both)
    rl, _ = C.loss_grad(kind, labels, logits, ll, tl, 0, want_grad=False)
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(l_only.cpu().numpy()), fin)
    assert np.abs(l_only.cpu().numpy()[fin] - rl[fin]).max() < TOL * max(1.0, np.abs(rl[fin]).max())
