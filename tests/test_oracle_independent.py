"""Cross-checks of the oracle against sources that are NOT the reference: brute-force path enumeration
(by the definitions in classic_ctc_loss.py:40-52 / simplified_ctc_loss.py:39-49), torch's CPU ctc_loss
(the role tf.nn.ctc_loss plays in tests/test_classic_ctc_loss.py:332-393), finite differences
(tests/finite_difference.py:89-112) and the invariants the reference asserts on random inputs.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ctc_oracle as O

KINDS = ["classic", "simplified"]


def _rand_case(rng, T, V, U, with_repeat=False):
    logits = rng.standard_normal((1, T, V))
    labels = rng.integers(1, V, (1, U))
    if with_repeat and U >= 2:
        labels[0, 1] = labels[0, 0]
    return logits, labels


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("T,V,U,rep", [(4, 3, 2, False), (5, 3, 2, True), (6, 3, 3, True), (5, 4, 1, False),
                                         (3, 3, 0, False)])
def test_brute_force(kind, T, V, U, rep):
    rng = np.random.default_rng(T * 100 + V * 10 + U)
    logits, labels = _rand_case(rng, T, V, U, rep)
    ll = np.array([U]); tl = np.array([T])
    lab = labels if U > 0 else np.zeros((1, 1), dtype=np.int64)
    data = O.ctc_loss(kind, lab, logits, ll, tl, 0)
    loss, post, pair = O.brute_force(kind, labels[0, :U], O.logit_to_logproba(logits)[0], 0)
    assert abs(data.loss[0] - loss) < 1e-10 or (np.isinf(loss) and np.isinf(data.loss[0]))
    if np.isinf(loss):
        assert np.all(data.gradient == 0) and np.all(data.hessian == 0)
        return
    assert np.abs(-data.gradient[0] - post).max() < 1e-10
    # H[t1,k1,t2,k2] = -P12/P + g g   
    g = data.gradient[0]
    expect = -pair + g[:, :, None, None] * g[None, None, :, :]
    assert np.abs(data.hessian[0] - expect).max() < 1e-9
    # symmetry, tests/test_hessian.py:89-108
    assert np.abs(data.hessian - np.transpose(data.hessian, (0, 3, 4, 1, 2))).max() < 1e-12


@pytest.mark.parametrize("B,T,V", [(8, 20, 8), (8, 64, 10)])
def test_classic_vs_torch_ctc(B, T, V):
    """Loss 5 places at B=8,T=20,V=8 and logits-gradient 4 places at B=8,T=64,V=10 in the reference
    (tests/test_classic_ctc_loss.py:332-393); in fp64 both agree to 1e-9."""
    inp = O.generate_ctc_loss_inputs(B, T, 0, V)
    x = torch.tensor(inp["logits"], dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, 2).transpose(0, 1)
    U = int(inp["label_length"].max())
    loss = torch.nn.functional.ctc_loss(lp, torch.tensor(inp["labels"][:, :U].astype(np.int64)),
                                        torch.tensor(inp["logit_length"].astype(np.int64)),
                                        torch.tensor(inp["label_length"].astype(np.int64)),
                                        blank=0, reduction="none", zero_infinity=False)
    loss.sum().backward()
    data = O.ctc_loss("classic", inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    assert np.abs(data.loss - loss.detach().numpy()).max() < 1e-9
    assert np.abs(O.logits_gradient(data, inp["logits"]) - x.grad.numpy()).max() < 1e-9


@pytest.mark.parametrize("kind", KINDS)
def test_alpha_beta_sum_invariant(kind):
    """sum_states exp(alpha+beta) = exp(-loss) for every t (tests/test_classic_ctc_loss.py:146-167,
    tests/test_simplified_ctc_loss.py:185-206)."""
    inp = O.generate_ctc_loss_inputs(3, 12, 1, 5)
    data = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    axes = (2, 3) if kind == "classic" else (2,)
    sums = O.reduce_logsumexp(data.alpha + data.beta, axis=axes)
    assert np.abs(sums + data.loss[:, None]).max() < 1e-10


def _fd_jacobian(f, x, eps):
    """tests/finite_difference.py:89-112 but central differences: d f[b,...] / d x[b,t,k]."""
    base = f(x)
    B, T, V = x.shape
    out = np.zeros(base.shape + (T, V))
    for t in range(T):
        for k in range(V):
            xp = x.copy(); xp[:, t, k] += eps
            xm = x.copy(); xm[:, t, k] -= eps
            out[..., t, k] = (f(xp) - f(xm)) / (2 * eps)
    return out


@pytest.mark.parametrize("kind", KINDS)
def test_gradient_and_hessian_vs_finite_differences(kind):
    """Gradient vs FD (2 places in tests/test_classic_ctc_loss.py:395-425), logits-Hessian vs FD
    (2 places at B=2,T=4,V=2, tests/test_hessian.py:149-183); central differences in fp64 reach 1e-7."""
    inp = O.generate_ctc_loss_inputs(2, 4, 0, 3, max_label_length=2)
    inp["logit_length"][:] = [4, 3]  # feasible for both variants (label_length <= 2, one repeat at most)
    inp["label_length"][:] = [2, 1]
    x = inp["logits"].astype(np.float64)
    args = (inp["labels"], inp["label_length"], inp["logit_length"], 0)

    def loss_fn(x_):
        return O.ctc_loss(kind, inp["labels"], x_, *args[1:]).loss

    def grad_fn(x_):
        d = O.ctc_loss(kind, inp["labels"], x_, *args[1:])
        return O.logits_gradient(d, x_)

    data = O.ctc_loss(kind, inp["labels"], x, *args[1:])
    g_fd = _fd_jacobian(loss_fn, x, 1e-6)  # [B, T, V]
    assert np.abs(O.logits_gradient(data, x) - g_fd).max() < 1e-7
    h_fd = _fd_jacobian(grad_fn, x, 1e-6)  # [B, T, V, T, V]
    assert np.abs(O.logits_hessian(data, x) - h_fd).max() < 1e-6
    # log-probability space Hessian: perturb log-probabilities directly (treated as independent variables)
    lp = O.logit_to_logproba(x)

    def lp_grad_fn(lp_):
        return O.LOSS_DATA[kind](inp["labels"], lp_, *args[1:]).gradient

    def lp_loss_fn(lp_):
        return O.LOSS_DATA[kind](inp["labels"], lp_, *args[1:]).loss

    assert np.abs(data.gradient - _fd_jacobian(lp_loss_fn, lp, 1e-6)).max() < 1e-7
    assert np.abs(data.hessian - _fd_jacobian(lp_grad_fn, lp, 1e-6)).max() < 1e-6


@pytest.mark.parametrize("kind", KINDS)
def test_padded_frames_and_inf_samples_are_zero(kind):
    inp = O.generate_ctc_loss_inputs(4, 8, 3, 4)
    inp["label_length"][1] = 7; inp["logit_length"][1] = 4  # infeasible sample
    data = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    assert np.isinf(data.loss[1]) and np.all(data.gradient[1] == 0) and np.all(data.hessian[1] == 0)
    for b in range(4):
        n = inp["logit_length"][b]
        assert np.all(data.gradient[b, n:] == 0)
        assert np.all(data.hessian[b, n:] == 0) and np.all(data.hessian[b, :, :, n:] == 0)
        if np.isfinite(data.loss[b]):
            assert np.abs(data.gradient[b, :n].sum(axis=1) + 1).max() < 1e-12
