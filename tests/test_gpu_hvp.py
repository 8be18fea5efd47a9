"""Hessian-vector products (ctc_amd_hvp, tangent-mode alpha/beta) against the materialised Hessian.

The reference obtains the second-order product by contracting its [B,T,V,T,V] Hessian inside gradient_fn.backprop
(base_loss.py:157-175); these tests check that the tangent-mode kernel gives the same numbers as
  * einsum(oracle Hessian, v)            (NumPy restatement of base_loss.py:186-260 / README.md:58-71), small sizes,
  * einsum(GPU dense Hessian, v)         (ctc_amd_hessian), sizes across the lane tilings,
  * central differences of the gradient  at the north-star size, where no Hessian fits in memory,
and that autograd's double backward takes this route.  Tolerance 1e-4 (relative to max|Hv|, floor 1), float32.
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


def _prep(inp, blank=0):
    from tf_seq2seq_losses_amd import ops
    return ops.Prepared(_t(inp["labels"]), _t(inp["logits"]), _t(inp["label_length"]), _t(inp["logit_length"]), blank)


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _fd64(kind, inp, v, eps):
    """Central differences of the float64 C-oracle gradient.  The perturbed logits are rounded to float32 first and the
    direction actually taken, (x+ - x-) / 2 eps, is returned with the quotient, so rounding adds no error."""
    xp = (inp["logits"].astype(np.float64) + eps * v).astype(np.float32)
    xm = (inp["logits"].astype(np.float64) - eps * v).astype(np.float32)
    veff = (xp.astype(np.float64) - xm.astype(np.float64)) / (2 * eps)
    gp = C.loss_grad(kind, inp["labels"], xp, inp["label_length"], inp["logit_length"], 0)[1]
    gm = C.loss_grad(kind, inp["labels"], xm, inp["label_length"], inp["logit_length"], 0)[1]
    return (gp - gm) / (2 * eps), veff.astype(np.float32)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,V,seed", [(8, 20, 8, 0), (3, 12, 5, 2), (2, 4, 2, 3), (5, 33, 7, 4)])
def test_hvp_matches_oracle_hessian(kind, B, T, V, seed):
    from tf_seq2seq_losses_amd import ops, _lib
    inp = O.generate_ctc_loss_inputs(B, T, seed, V)
    rng = np.random.default_rng(100 + seed)
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    # logits space: README.md:58-71
    ref = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    want = np.einsum("btkuj,buj->btk", O.logits_hessian(ref, inp["logits"]), v.astype(np.float64))
    loss, grad, out = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, _prep(inp), _t(v), want_grad=True)
    assert _rel(out.cpu().numpy(), want) < TOL
    assert np.abs(grad.cpu().numpy() - O.logits_gradient(ref, inp["logits"])).max() < TOL
    fin = np.isfinite(ref.loss)
    assert np.array_equal(np.isfinite(loss.cpu().numpy()), fin)
    # log-probability space: loss_data.hessian, base_loss.py:186-260
    import tf_seq2seq_losses_amd as ctc
    lp = torch.log_softmax(_t(inp["logits"]), dim=2)
    cls = ctc.ClassicCtcLossData if kind == "classic" else ctc.SimplifiedCtcLossData
    data = cls(_t(inp["labels"]), lp, _t(inp["label_length"]), _t(inp["logit_length"]), 0)
    refd = O.LOSS_DATA[kind](inp["labels"], lp.cpu().numpy(), inp["label_length"], inp["logit_length"], 0)
    want_lp = np.einsum("btkuj,buj->btk", refd.hessian, v.astype(np.float64))
    assert _rel(data.hessian_vector_product(_t(v)).cpu().numpy(), want_lp) < TOL


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("T,V,U", [(40, 12, 17), (150, 9, 70), (200, 6, 130), (300, 4, 260)])
def test_hvp_across_lane_tilings(kind, T, V, U):
    """NL = 1, 2, 4, 8 label positions per lane, repeats, ragged lengths, one empty label, one infeasible sample.
    Checked against float64 central differences of the C oracle's gradient (2e-4: the differences themselves carry
    ~4e-5 of truncation) and, loosely (1e-3), against the contraction of the float32 dense Hessian kernel, whose own
    error at T = 200 is ~5e-4 (it sums T*V float32 entries of g (x) g - P12 per output)."""
    from tf_seq2seq_losses_amd import ops, _lib
    inp = O.generate_ctc_loss_inputs(4, T, U, V, max_label_length=U)
    inp["labels"][0, : U // 2] = 1
    inp["label_length"][1] = 0
    inp["logit_length"][2] = max(1, int(inp["label_length"][2]) - 1)  # shorter than the label: infeasible
    rng = np.random.default_rng(U)
    fd, v = _fd64(kind, inp, rng.standard_normal((4, T, V)), 1e-3)
    feas = [0, 1, 3]
    p = _prep(inp)
    loss, _, out = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, p, _t(v))
    outn = out.cpu().numpy()
    assert np.isfinite(outn).all()
    assert _rel(outn[feas], fd[feas]) < 2e-4, _rel(outn[feas], fd[feas])
    assert loss[2].item() == float("inf") and np.all(outn[2] == 0)
    assert np.all(outn[0, int(inp["logit_length"][0]):] == 0)
    for wrt in (_lib.WRT_LOGITS, _lib.WRT_LOGPROBS):
        if wrt == _lib.WRT_LOGPROBS:
            p = ops.Prepared(_t(inp["labels"]), torch.log_softmax(_t(inp["logits"]), 2), _t(inp["label_length"]),
                             _t(inp["logit_length"]), 0)
            out = ops.hvp(ops.KINDS[kind], wrt, p, _t(v))[2]
        _, _, hess = ops.hessian(ops.KINDS[kind], wrt, p, want_grad=False)
        want = torch.einsum("btkuj,buj->btk", hess.double(), _t(v).double()).cpu().numpy()
        del hess
        assert _rel(out.cpu().numpy(), want) < 1e-3, (wrt, _rel(out.cpu().numpy(), want))


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_hvp_nonzero_blank_and_symmetry(kind):
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(5)
    B, T, V, U = 3, 30, 11, 9
    inp = dict(logits=rng.standard_normal((B, T, V)).astype(np.float32) * 3,
               labels=rng.integers(0, V - 1, (B, U)).astype(np.int32),
               label_length=np.array([9, 4, 7], np.int32), logit_length=np.array([30, 22, 15], np.int32))
    blank = V - 1
    u, v = (rng.standard_normal((B, T, V)).astype(np.float32) for _ in range(2))
    p = _prep(inp, blank)
    hu = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, p, _t(u))[2].double()
    hv = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, p, _t(v))[2].double()
    a, b = (_t(v).double() * hu).sum((1, 2)), (_t(u).double() * hv).sum((1, 2))
    assert ((a - b).abs() / (1 + a.abs())).max().item() < TOL          # <v, Hu> = <u, Hv>
    assert hv.sum(2).abs().max().item() < TOL                          # logits-space rows sum to zero (softmax gauge)
    _, _, hess = ops.hessian(ops.KINDS[kind], _lib.WRT_LOGITS, p, want_grad=False)
    want = torch.einsum("btkuj,buj->btk", hess.double(), _t(v).double())
    assert ((hv - want).abs().max() / max(1.0, want.abs().max().item())).item() < TOL


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_autograd_second_order_uses_hvp(kind, monkeypatch):
    """README.md:58-71 route: grad of <gradient, v>; the materialised route (losses.HVP_DENSE) must agree."""
    import tf_seq2seq_losses_amd as ctc
    inp = O.generate_ctc_loss_inputs(4, 24, 7, 6)
    fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simplified_ctc_loss
    v = _t(np.random.default_rng(3).standard_normal(inp["logits"].shape).astype(np.float32))

    def second(weights):
        x = _t(inp["logits"]).requires_grad_(True)
        loss = fn(_t(inp["labels"]), x, _t(inp["label_length"]), _t(inp["logit_length"]), 0)
        fin = torch.isfinite(loss)
        (g,) = torch.autograd.grad((loss[fin] * weights[fin]).sum(), x, create_graph=True)
        (h,) = torch.autograd.grad((g * v).sum(), x)
        return h

    wts = _t(np.array([1.0, 2.0, -0.5, 3.0], np.float32))
    fast = second(wts)
    from tf_seq2seq_losses_amd import losses
    monkeypatch.setattr(losses, "HVP_DENSE", True)
    dense = second(wts)
    assert torch.isfinite(fast).all()
    assert ((fast - dense).abs().max() / max(1.0, dense.abs().max().item())).item() < TOL


def _fd64_exact(kind, inp, v, eps=2e-3):
    """Directional derivative of the float64 NumPy oracle's gradient along v, by central differences at eps and 2 eps with
    Richardson extrapolation (error O(eps^4)); logits and direction are taken in float64 exactly as the GPU sees them."""
    x = inp["logits"].astype(np.float64)
    vv = v.astype(np.float64)

    def grad(z):
        d = O.ctc_loss(kind, inp["labels"], z, inp["label_length"], inp["logit_length"], 0)
        return O.logits_gradient(d, z)
    d1 = (grad(x + eps * vv) - grad(x - eps * vv)) / (2 * eps)
    d2 = (grad(x + 2 * eps * vv) - grad(x - 2 * eps * vv)) / (4 * eps)
    return (4.0 * d1 - d2) / 3.0


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_hvp_north_star_size_finite_differences(kind):
    """B=4 of the north-star shape (T=1000, U=128, V=256): the dense Hessian would need 262 GB per utterance, so the check is
    against the directional derivative of the float64 oracle's gradient (Richardson-extrapolated central differences: error far
    below 1e-6 of max|Hv|).  The fused linear-domain kernel (ctc_hvp_fused.hip) is held to north_star's 1e-4 of max|Hv|; the
    log-domain pipeline it replaced at this shape (still the route of flagged utterances, forced here through the override)
    measured 2.3e-4 / 2.9e-4 in r02; since r03 it takes posteriors and their tangents relative to the frame's own mass
    (ctc_hvp_device.h) and is held to the same 1e-4.  Symmetry <u,Hv> = <v,Hu> on the scale |v| |Hu|: 2e-6."""
    from tf_seq2seq_losses_amd import ops, _lib
    B, T, U, V = 4, 1000, 128, 256
    rng = np.random.default_rng(11)
    inp = dict(logits=rng.standard_normal((B, T, V)).astype(np.float32),
               labels=rng.integers(1, V, (B, U)).astype(np.int32),
               label_length=np.array([128, 100, 64, 128], np.int32), logit_length=np.array([1000, 900, 1000, 517], np.int32))
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    fd = _fd64_exact(kind, inp, v)
    out = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, _prep(inp), _t(v))[2]
    assert torch.isfinite(out).all()
    outn = out.cpu().numpy().astype(np.float64)
    for b in range(B):
        assert np.abs(outn[b] - fd[b]).max() < 1e-4 * np.abs(fd[b]).max(), (b, np.abs(outn[b] - fd[b]).max() / np.abs(fd[b]).max())
        assert np.all(outn[b, int(inp["logit_length"][b]):] == 0)
    _lib.debug_override("hvp", "v1")
    try:
        old = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, _prep(inp), _t(v))[2].cpu().numpy().astype(np.float64)
    finally:
        _lib.debug_override("hvp", "")
    for b in range(B):
        e = np.abs(old[b] - fd[b]).max() / np.abs(fd[b]).max()
        print(f"log-domain HVP pipeline, {kind}, utterance {b}: {e:.2e} of max|Hv|")
        assert e < 1e-4, (b, e)
    # <u, Hv> = <v, Hu> at this size (256k-term inner products: compared on the scale |v| |Hu|)
    u = rng.standard_normal((B, T, V)).astype(np.float32)
    hu = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, _prep(inp), _t(u))[2].double()
    a, b_ = (_t(v).double() * hu).sum((1, 2)), (_t(u).double() * out.double()).sum((1, 2))
    scale = _t(v).double().flatten(1).norm(dim=1) * hu.flatten(1).norm(dim=1)
    assert ((a - b_).abs() / scale).max().item() < 2e-6


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,V,U,blank", [(5, 37, 8, 7, 0), (6, 200, 64, 32, 0), (3, 150, 256, 100, 255), (4, 90, 28, 64, 3), (2, 300, 256, 128, 0)])
def test_fused_hvp_against_the_oracle_and_the_log_domain_pipeline(kind, B, T, V, U, blank):
    """The fused kernel on every instantiation (one / two label positions per lane, masked vocabularies, a non-zero blank,
    ragged lengths, an empty label, an infeasible sample): 1e-4 of max|Hv| against the float64 directional derivative, the
    log-domain pipeline agrees, zero rows beyond logit_length, +inf / zeros for the infeasible sample."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(B * 1000 + T)
    tok = np.array([k for k in range(V) if k != blank])
    inp = dict(logits=rng.standard_normal((B, T, V)).astype(np.float32),
               labels=tok[rng.integers(0, V - 1, (B, U))].astype(np.int32),
               label_length=rng.integers(max(1, U // 2), U + 1, B).astype(np.int32),
               logit_length=rng.integers(max(U + 2, T // 2), T + 1, B).astype(np.int32))
    inp["labels"][0, : U // 2] = inp["labels"][0, 0]            # repeats (classic: needs blanks in between)
    inp["logit_length"][0] = T
    inp["label_length"][1] = 0                                   # empty label
    if B > 2:
        inp["logit_length"][2] = max(1, int(inp["label_length"][2]) - 1)   # infeasible
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    p = ops.Prepared(_t(inp["labels"]), _t(inp["logits"]), _t(inp["label_length"]), _t(inp["logit_length"]), blank)
    k = ops.KINDS[kind]
    loss, _, out = ops.hvp(k, _lib.WRT_LOGITS, p, _t(v))
    _lib.debug_override("hvp", "v1")
    try:
        loss1, _, out1 = ops.hvp(k, _lib.WRT_LOGITS, p, _t(v))
    finally:
        _lib.debug_override("hvp", "")
    assert torch.isfinite(out).all()
    rl = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], blank).loss
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(loss.cpu().numpy()), fin) and np.array_equal(np.isfinite(loss1.cpu().numpy()), fin)
    assert (np.abs(loss.cpu().numpy()[fin] - rl[fin]) / np.maximum(1.0, np.abs(rl[fin]))).max() < TOL

    def fd_blank(z):
        d = O.ctc_loss(kind, inp["labels"], z, inp["label_length"], inp["logit_length"], blank)
        g = O.logits_gradient(d, z)
        return np.where(fin[:, None, None], g, 0.0)
    x, vv, eps = inp["logits"].astype(np.float64), v.astype(np.float64), 2e-3
    d1 = (fd_blank(x + eps * vv) - fd_blank(x - eps * vv)) / (2 * eps)
    d2 = (fd_blank(x + 2 * eps * vv) - fd_blank(x - 2 * eps * vv)) / (4 * eps)
    fd = (4.0 * d1 - d2) / 3.0
    outn, out1n = out.cpu().numpy().astype(np.float64), out1.cpu().numpy().astype(np.float64)
    for b in range(B):
        scale = max(1e-3, np.abs(fd[b]).max())
        assert np.abs(outn[b] - fd[b]).max() < TOL * max(1.0, scale), (b, np.abs(outn[b] - fd[b]).max(), scale)
        assert np.abs(out1n[b] - fd[b]).max() < TOL * max(1.0, scale), b
        assert np.all(outn[b, int(inp["logit_length"][b]):] == 0)
        if not fin[b]:
            assert np.all(outn[b] == 0)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_fused_hvp_hands_flagged_utterances_to_the_log_domain_pipeline(kind):
    """Sharp logits (N(0, 6^2), 1e10, -inf columns) leave the linear-domain format: the fused kernel flags those utterances and
    the log-domain pipeline, restricted to them, writes their rows (the same numbers as a call forced onto that pipeline, up to
    the rounding of its one-frame emission kernel against the four-frame one), while the benign utterances of the same batch
    keep the fused kernel's result."""
    from tf_seq2seq_losses_amd import ops, _lib
    rng = np.random.default_rng(9)
    B, T, V, U = 6, 400, 64, 40
    logits = (rng.standard_normal((B, T, V)) * np.array([1, 6, 1, 6, 1, 1])[:, None, None]).astype(np.float32)
    logits[4, :, 5] = -np.inf
    logits[5, 7, 1] = 1e10
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    labels[4][labels[4] == 5] = 6
    inp = dict(logits=logits, labels=labels, label_length=np.full(B, U, np.int32), logit_length=np.full(B, T, np.int32))
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    k = ops.KINDS[kind]
    loss, _, out = ops.hvp(k, _lib.WRT_LOGITS, _prep(inp), _t(v))
    _lib.debug_override("hvp", "v1")
    try:
        loss1, _, out1 = ops.hvp(k, _lib.WRT_LOGITS, _prep(inp), _t(v))
    finally:
        _lib.debug_override("hvp", "")
    assert torch.isfinite(out).all()
    fin = torch.isfinite(loss1)
    assert torch.equal(torch.isfinite(loss), fin)
    assert ((loss[fin] - loss1[fin]).abs() / loss1[fin].abs()).max().item() < 1e-5
    for b in range(B):  # whichever kernel wrote an utterance's rows, they are the product (the 1e10 utterance would be NaN otherwise)
        assert ((out[b] - out1[b]).abs().max() / max(1e-6, out1[b].abs().max().item())).item() < 5e-4, b
    # benign utterances against the float64 directional derivative: the fused kernel's accuracy, not the log-domain one's
    sub = {k2: v2[[0, 2]] for k2, v2 in inp.items()}
    fd = _fd64_exact(kind, sub, v[[0, 2]])
    got = out[[0, 2]].cpu().numpy().astype(np.float64)
    for i in range(2):
        assert np.abs(got[i] - fd[i]).max() < TOL * np.abs(fd[i]).max(), i
