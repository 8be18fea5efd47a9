"""BASELINE.json configs that round 1 only ran at reduced size, now at full size, plus the generic entry points.

* configs[4]: dense Hessian at B=32 T=200 U=32 V=64 -- the full batch (21 GB output), both Hessian kernels.  The full-B
  launch goes through hess_plan_kernel and the XCD stride permutation, a different work order from the B=2 case of
  test_gpu_large.py.  Checked: finite, zero rows/columns past logit_length, softmax gauge sum_j H = 0, symmetry (per
  utterance, no 21 GB transpose), and for 4 utterances spread over the batch H.v against ctc_amd_hvp and against float64
  central differences of the C oracle's gradient.
* configs[3]: the per-rank slices of the B=2048 batch are the configs[1] tensor drawn with seeds 0..7 (bench.py seeds by
  rank); seeds 0/1 are in test_gpu_large.py, seeds 2..7 here: 8 utterances each against the float64 C oracle.
* ctc_loss / ctc_loss_from_logproba (base_loss.py:38-99) with first- and second-order autograd (base_loss.py:140-184).
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import ctc_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _inputs(B, T, U, V, seed):
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((B, T, V), dtype=np.float32)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    return logits, labels, np.full(B, U, dtype=np.int32), np.full(B, T, dtype=np.int32)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("kernel", ["pair", "slab"])
def test_config5_full_batch_hessian(kind, kernel):
    from tf_seq2seq_losses_amd import ops, _lib
    _lib.debug_override("hessian", "slab" if kernel == "slab" else "")
    try:
        _config5_body(kind)
    finally:
        _lib.debug_override("hessian", "")


def _golden_slabs():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_hessian_slabs.npz"))


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("kernel", ["pair", "slab"])
def test_config5_logprob_space_hessian_entries(kind, kernel):
    """loss_data.hessian (base_loss.py:186-260, derivatives w.r.t. log-probabilities) at the full T=200 U=32 V=64 of
    configs[4] for utterances 0 and 13, entry by entry against the float64 gamma-oracle's slabs, 1e-4."""
    import tf_seq2seq_losses_amd as ctc
    from tf_seq2seq_losses_amd import _lib
    B, T, U, V = 32, 200, 32, 64
    logits, labels, ll, tl = _inputs(B, T, U, V, 0)
    pick = [0, 13]
    lp = torch.log_softmax(_t(logits[pick]).double(), dim=2).float()
    cls = ctc.ClassicCtcLossData if kind == "classic" else ctc.SimplifiedCtcLossData
    gold = _golden_slabs()
    _lib.debug_override("hessian", "slab" if kernel == "slab" else "")
    try:
        h = cls(_t(labels[pick]), lp, _t(ll[pick]), _t(tl[pick]), 0).hessian
    finally:
        _lib.debug_override("hessian", "")
    for i, b in enumerate(pick):
        idx = gold[f"u{b}/index"]
        want = gold[f"u{b}/{kind}/logprobs"]
        got = torch.stack([h[i, int(t1), int(k1)] for t1, k1 in idx]).cpu().numpy()
        assert np.abs(got - want).max() < TOL, (b, np.abs(got - want).max())


def _config5_body(kind):
    from tf_seq2seq_losses_amd import ops, _lib
    B, T, U, V = 32, 200, 32, 64
    logits, labels, ll, tl = _inputs(B, T, U, V, 0)
    tl[5], tl[30] = 150, 97          # two shorter utterances: rows and columns past the end must be zero
    ll[11] = 20
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0)
    k = ops.KINDS[kind]
    loss, grad, h = ops.hessian(k, _lib.WRT_LOGITS, p)
    assert torch.isfinite(loss).all() and torch.isfinite(grad).all()
    rl, rg = C.loss_grad(kind, labels, logits, ll, tl, 0)
    assert (np.abs(loss.cpu().numpy() - rl) / np.abs(rl)).max() < TOL
    assert np.abs(grad.cpu().numpy() - rg).max() < TOL
    worst_sym = worst_gauge = 0.0
    for b in range(B):                                   # per utterance: 655 MB slices, nothing bigger is formed
        hb = h[b].reshape(T * V, T * V)
        assert torch.isfinite(hb).all()
        worst_sym = max(worst_sym, (hb - hb.t()).abs().max().item())
        worst_gauge = max(worst_gauge, h[b].sum(dim=3).abs().max().item())
        n = int(tl[b])
        if n < T:
            assert h[b, n:].abs().max().item() == 0 and h[b, :, :, n:].abs().max().item() == 0
    # entry by entry against the float64 gamma-oracle at FULL size: 20 slabs H[b, t1, k1, :, :] of utterances 0 and 13
    # (tests/golden/config5_hessian_slabs.npz, written by tests/golden/make_config5_slabs.py from oracle/ctc_oracle.py)
    gold = _golden_slabs()
    for b in (0, 13):
        idx = gold[f"u{b}/index"]
        want = gold[f"u{b}/{kind}/logits"]
        got = torch.stack([h[b, int(t1), int(k1)] for t1, k1 in idx]).cpu().numpy()
        assert np.abs(got - want).max() < TOL, (b, np.abs(got - want).max())
        assert abs(float(loss[b]) - float(gold[f"u{b}/{kind}/loss"][0])) < TOL * float(gold[f"u{b}/{kind}/loss"][0])
    # the two triangles are generated independently (forward / backward propagation): float32 rounding of T = 200 recursions
    assert worst_sym < 1e-4, worst_sym
    assert worst_gauge < 1e-4, worst_gauge
    # H.v for utterances spread over the batch (different XCDs / positions of the stride permutation)
    pick = [0, 5, 13, 31]
    rng = np.random.default_rng(3)
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    for b in pick:
        v[b, tl[b]:] = 0
    vt = _t(v)
    _, _, hv_kernel = ops.hvp(k, _lib.WRT_LOGITS, p, vt)
    eps = 2e-3   # float64 oracle gradients: the step is limited only by float32 rounding of x +- eps v (veff below)
    sub = dict(labels=labels[pick], logits=logits[pick], label_length=ll[pick], logit_length=tl[pick])
    xp = (sub["logits"].astype(np.float64) + eps * v[pick]).astype(np.float32)
    xm = (sub["logits"].astype(np.float64) - eps * v[pick]).astype(np.float32)
    veff = ((xp.astype(np.float64) - xm.astype(np.float64)) / (2 * eps)).astype(np.float32)
    gp = C.loss_grad(kind, sub["labels"], xp, sub["label_length"], sub["logit_length"], 0)[1]
    gm = C.loss_grad(kind, sub["labels"], xm, sub["label_length"], sub["logit_length"], 0)[1]
    fd = (gp - gm) / (2 * eps)
    for i, b in enumerate(pick):
        hb = h[b].reshape(T * V, T * V)
        dense = (hb @ _t(veff[i]).reshape(-1)).reshape(T, V).cpu().numpy()
        scale = max(1.0, np.abs(fd[i]).max())
        # dense float32 Hessian contracted with v: sums T*V float32 entries per output (measured ~5e-4 at T = 200)
        assert np.abs(dense - fd[i]).max() / scale < 1e-3, (b, np.abs(dense - fd[i]).max(), scale)
        tang = hv_kernel[b].cpu().numpy()
        want = (hb @ vt[b].reshape(-1)).reshape(T, V).cpu().numpy()
        assert np.abs(tang - want).max() / max(1.0, np.abs(want).max()) < 1e-3, b


@pytest.mark.parametrize("seed", [2, 3, 4, 5, 6, 7])
def test_config4_rank_slices(seed):
    """Rank r of the 8-GPU run processes the configs[1] tensor drawn with seed r (bench.py make_inputs).  The whole
    256-utterance slice runs on the GPU; the first 8 utterances are compared with the float64 C oracle."""
    from tf_seq2seq_losses_amd import ops, _lib
    B, T, U, V = 256, 1000, 128, 256
    logits, labels, ll, tl = _inputs(B, T, U, V, seed)
    p = ops.Prepared(_t(labels), _t(logits), _t(ll), _t(tl), 0, U=U)
    loss, grad = ops.loss_grad(_lib.CLASSIC, _lib.WRT_LOGITS, p, True)
    lossn = loss.cpu().numpy()
    assert np.isfinite(lossn).all() and torch.isfinite(grad).all()
    assert grad.sum(dim=2).abs().max().item() < TOL           # softmax - posterior sums to zero in every frame
    n = 8
    rl, rg = C.loss_grad("classic", labels[:n], logits[:n], ll[:n], tl[:n], 0)
    assert (np.abs(lossn[:n] - rl) / np.abs(rl)).max() < TOL
    err = np.abs(grad[:n].cpu().numpy() - rg).max()
    assert err < TOL, err


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("entry", ["ctc_loss", "ctc_loss_from_logproba"])
def test_generic_entry_points_autograd(kind, entry):
    """ctc_loss (base_loss.py:38-68: derivatives w.r.t. logits) and ctc_loss_from_logproba (base_loss.py:71-99:
    log-probabilities as independent variables), first order with d_loss weights and second order as a
    Hessian-vector product, against the NumPy oracle; third order refused."""
    import tf_seq2seq_losses_amd as ctc
    cls = ctc.ClassicCtcLossData if kind == "classic" else ctc.SimplifiedCtcLossData
    inp = O.generate_ctc_loss_inputs(4, 11, 21, 5, max_label_length=4)
    inp["logit_length"][:] = [11, 7, 9, 11]
    w = torch.tensor([0.5, -2.0, 3.0, 1.0], device=_dev())
    args = (_t(inp["labels"]), _t(inp["label_length"]), _t(inp["logit_length"]))
    if entry == "ctc_loss":
        x = _t(inp["logits"]).requires_grad_(True)
        loss = ctc.ctc_loss(args[0], x, args[1], args[2], 0, cls)
        ref = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
        gref = O.logits_gradient(ref, inp["logits"], d_loss=w.cpu().numpy())
        href = O.logits_hessian(ref, inp["logits"])
    else:
        lp = torch.log_softmax(_t(inp["logits"]), dim=2)
        x = lp.detach().clone().requires_grad_(True)
        loss = ctc.ctc_loss_from_logproba(args[0], x, args[1], args[2], 0, cls)
        ref = O.LOSS_DATA[kind](inp["labels"], lp.cpu().numpy(), inp["label_length"], inp["logit_length"], 0)
        gref = ref.gradient * w.cpu().numpy()[:, None, None]
        href = ref.hessian
    lossn = loss.detach().cpu().numpy()
    fin = np.isfinite(ref.loss)
    assert np.array_equal(np.isfinite(lossn), fin)
    assert (np.abs(lossn[fin] - ref.loss[fin]) / np.maximum(1, np.abs(ref.loss[fin]))).max() < TOL
    (g,) = torch.autograd.grad((loss * w).sum(), x, create_graph=True)
    assert np.abs(g.detach().cpu().numpy() - gref).max() < TOL
    v = torch.randn(g.shape, device=g.device, generator=torch.Generator(device=g.device).manual_seed(1))
    (hv,) = torch.autograd.grad((g * v).sum(), x, create_graph=True)
    want = np.einsum("btkuj,buj->btk", href, v.cpu().numpy().astype(np.float64)) * w.cpu().numpy()[:, None, None]
    assert np.abs(hv.detach().cpu().numpy() - want).max() < TOL
    with pytest.raises(NotImplementedError):
        torch.autograd.grad(hv.sum(), x)
    # the same loss through the named public function (classic_ctc_loss / simplified_ctc_loss call ctc_loss)
    if entry == "ctc_loss":
        fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simplified_ctc_loss
        assert torch.equal(fn(args[0], _t(inp["logits"]), args[1], args[2], 0), loss.detach())
