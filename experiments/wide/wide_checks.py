"""Checks of the EXPERIMENTAL one-launch tier for vocabularies beyond the fused kernels (csrc/ctc_wide.hip, pipeline "wide":
emission, chain and gradient stages beside each other in one persistent grid, rows handed over through flags) against the
float64 C oracle and against the three kernels.  The tier is not part of the product library: build the
diagnostic library first and point the binding at it --
    bash scripts/build_wide_variant.sh && CTC_AMD_LIB=scratch/libctc_wide_diag.so python -m pytest tests/tools/wide_checks.py -q
(not collected by `pytest tests`: the file name does not match test_*.py).

Edge cases the reference tests (tests/test_ctc_losses.py, tests/test_classic_ctc_loss.py): empty label, label longer than the
frames allow (loss +inf, zero gradient), logit_length 0, ragged lengths; plus what is specific to this kernel: frame counts
that are not a multiple of the 4-frame task or of the 64-frame flag chunk, an odd batch (half-empty chain workgroup), every
label tier (1, 2, 4 positions per lane; longer labels run the three kernels), a batch larger than one launch takes, and stream capture.
"""
import numpy as np
import pytest
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: loss / gradient within 1e-4 of the reference (float64 oracle here)


def _dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _run(kind, logits, labels, ll, tl, blank=0, pipeline="", d_loss=None, U=None):
    from tf_seq2seq_losses_amd import ops, _lib
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), blank, U=U if U is not None else labels.shape[1])
    _lib.debug_override("pipeline", pipeline)
    try:
        name = ops.pipeline_of(ops.KINDS[kind], _lib.WRT_LOGITS, p)
        loss, grad = ops.loss_grad(ops.KINDS[kind], _lib.WRT_LOGITS, p, True, d_loss=None if d_loss is None else _t(d_loss))
        torch.cuda.synchronize()
    finally:
        _lib.debug_override("pipeline", "")
    return name, loss.cpu().numpy(), grad.cpu().numpy()


def _check(kind, logits, labels, ll, tl, blank=0, d_loss=None):
    name, loss, grad = _run(kind, logits, labels, ll, tl, blank, "wide", d_loss)
    # labels of more than 256 positions stay on the three-kernel pipeline (the LDS rings of a chain workgroup hold 8..16 rows)
    assert name == ("wide" if labels.shape[1] <= 256 else "v1")
    _, loss1, grad1 = _run(kind, logits, labels, ll, tl, blank, "v1", d_loss)
    rl, rg = C.loss_grad(kind, labels, logits, ll, tl, blank)
    if d_loss is not None:
        rg = rg * d_loss[:, None, None]
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(loss), fin), (loss, rl)
    assert np.all(loss[~fin] == np.inf)
    assert np.array_equal(loss, loss1)  # the same recursion, instruction for instruction
    if fin.any():
        assert (np.abs(loss[fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)).max() < 1e-5
    assert np.isfinite(grad).all()
    assert np.abs(grad - rg).max() < TOL, np.abs(grad - rg).max()
    assert np.abs(grad - grad1).max() < 2 * TOL  # (each within TOL of the oracle)
    assert not grad[~fin].any()
    for b in range(len(tl)):
        assert not grad[b, tl[b]:].any()
    return loss, grad


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,U,V", [(1, 1, 1, 1028), (2, 3, 2, 2048), (3, 64, 10, 1100), (5, 65, 20, 4096), (4, 129, 64, 2052),
                                     (7, 200, 65, 1280), (3, 131, 128, 3000), (2, 260, 129, 1536), (2, 300, 257, 1028),
                                     (2, 700, 520, 1040), (9, 17, 0, 1284)])
def test_wide_against_the_oracle(kind, B, T, U, V):
    rng = np.random.default_rng(B * 7 + T * 13 + U)
    logits = (rng.standard_normal((B, T, V)) * rng.choice([0.5, 1.0, 3.0])).astype(np.float32)
    labels = rng.integers(1, V, (B, max(U, 1))).astype(np.int32)
    if U >= 4:
        labels[0, : U // 2] = labels[0, 0]  # a run of repeats
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    tl = rng.integers(0, T + 1, B).astype(np.int32)
    ll[0], tl[0] = U, T
    if B > 2:
        tl[1] = 0                      # an utterance without frames
        ll[2], tl[2] = U, max(U - 1, 0)  # more labels than frames: infeasible unless U == 0
    _check(kind, logits, labels, ll, tl, 0, d_loss=rng.standard_normal(B).astype(np.float32) if B % 2 else None)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_wide_blank_in_a_later_pass_and_label_length_beyond_U(kind):
    B, T, U, V, blank = 4, 90, 21, 2560, 1500
    rng = np.random.default_rng(5)
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    labels[labels >= blank] += 1
    labels[1, ::2] = 2559
    ll = np.array([U, U - 2, 5, U], np.int32)
    tl = np.array([T, T - 4, T, T - 7], np.int32)
    loss, grad = _check(kind, logits, labels, ll, tl, blank)
    # contract violation (label_length > U): reported as infeasible, gradient zero, the other utterances untouched
    name, loss2, grad2 = _run(kind, logits, labels, np.array([U, U - 2, 5, U], np.int32), tl, blank, "wide", U=U - 1)
    assert name == "wide" and np.isinf(loss2[0]) and np.isinf(loss2[3]) and not grad2[0].any() and not grad2[3].any()
    assert np.array_equal(loss2[1:3], loss[1:3]) and np.array_equal(grad2[1:3], grad[1:3])


def test_wide_rows_sum_to_zero_and_match_v1_at_the_bench_size():
    """B=32 T=1000 U=128 V=4096 (bench.py's wide-vocabulary workload): every valid gradient row sums to zero (softmax and
    posterior both carry mass 1), padded rows are zero, and the result agrees with the three-kernel pipeline; a sample of
    utterances against the oracle."""
    B, T, U, V = 32, 1000, 128, 4096
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((B, T, V), dtype=np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    ll = rng.integers(U // 2, U + 1, B).astype(np.int32)
    tl = rng.integers(T // 2, T + 1, B).astype(np.int32)
    ll[0], tl[0] = U, T
    name, loss, grad = _run("classic", logits, labels, ll, tl, pipeline="wide")
    assert name == "wide"
    _, loss1, grad1 = _run("classic", logits, labels, ll, tl, pipeline="v1")
    assert np.array_equal(loss, loss1)
    sel = [0, 7, 31]
    rl, rg = C.loss_grad("classic", labels[sel], logits[sel], ll[sel], tl[sel], 0)
    assert (np.abs(loss[sel] - rl) / np.abs(rl)).max() < 1e-5
    err, err1 = np.abs(grad[sel] - rg).max(), np.abs(grad1[sel] - rg).max()
    print(f"gradient error vs float64 at T = 1000: wide {err:.2e}, three-kernel pipeline {err1:.2e}")
    assert err < TOL
    assert err1 < TOL and np.abs(grad - grad1).max() < 2 * TOL  # (each within TOL of the oracle)
    assert np.abs(grad.sum(axis=2)).max() < 2e-5
    for b in range(B):
        assert not grad[b, tl[b]:].any()


def test_wide_batch_larger_than_one_launch_and_repeatable():
    """More utterances than one launch takes (half of the resident grid may be chain workgroups): the host splits the
    batch; and two calls on the same workspace give the same bits (the flags are cleared per launch)."""
    from tf_seq2seq_losses_amd import ops, _lib
    B, T, U, V = 1100, 9, 3, 1028
    rng = np.random.default_rng(3)
    logits = rng.standard_normal((B, T, V), dtype=np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    tl = rng.integers(0, T + 1, B).astype(np.int32)
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
    k = ops.KINDS["classic"]
    ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, k, B, T, V, U), dtype=torch.uint8, device=_dev())
    ws.fill_(0xFF)  # nothing may rely on a zeroed workspace
    _lib.debug_override("pipeline", "wide")
    try:
        assert ops.pipeline_of(k, _lib.WRT_LOGITS, p) == "wide"
        loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
        loss2, grad2 = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
    finally:
        _lib.debug_override("pipeline", "")
    assert torch.equal(loss, loss2) and torch.equal(grad, grad2)
    rl, rg = C.loss_grad("classic", labels, logits, ll, tl, 0)
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(loss.cpu().numpy()), fin)
    assert np.abs(loss.cpu().numpy()[fin] - rl[fin]).max() < 1e-4
    assert np.abs(grad.cpu().numpy() - rg).max() < TOL


def test_wide_in_a_captured_graph():
    """The call (a memset node + one kernel node) captured into a hipGraph and replayed on new input values."""
    from tf_seq2seq_losses_amd import ops, _lib
    B, T, U, V = 6, 150, 30, 2048
    rng = np.random.default_rng(9)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    ll = rng.integers(1, U + 1, B).astype(np.int32)
    tl = rng.integers(T // 2, T + 1, B).astype(np.int32)
    x = torch.zeros((B, T, V), dtype=torch.float32, device=_dev())
    p = ops.Prepared(_t(labels), x, _t(ll), _t(tl), 0, U=U)
    k = ops.KINDS["classic"]
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, k, B, T, V, U), dtype=torch.uint8, device=_dev())
    _lib.debug_override("pipeline", "wide")
    try:
        assert ops.pipeline_of(k, _lib.WRT_LOGITS, p) == "wide"
        ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)  # (first call outside the capture: occupancy query)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
    finally:
        _lib.debug_override("pipeline", "")
    for seed in (1, 2):
        logits = np.random.default_rng(seed).standard_normal((B, T, V)).astype(np.float32)
        x.copy_(_t(logits))
        g.replay()
        torch.cuda.synchronize()
        rl, rg = C.loss_grad("classic", labels, logits, ll, tl, 0)
        assert (np.abs(loss.cpu().numpy() - rl) / np.abs(rl)).max() < 1e-5
        assert np.abs(grad.cpu().numpy() - rg).max() < TOL
