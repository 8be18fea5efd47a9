"""Golden fixtures (tests/golden/oracle_vectors.npz, self-generated -- see make_oracle_vectors.py).
CPU: the oracle still reproduces them.  GPU: the HIP path matches them within the north-star tolerance."""
import os

import numpy as np
import pytest

from oracle import ctc_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
NAMES = sorted({k.split("/")[0] for k in G.files})
TOL = 1e-4


def _inp(name):
    return {k: G[f"{name}/{k}"] for k in ("labels", "logits", "label_length", "logit_length")}


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name, kind):
    inp = _inp(name)
    d = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    np.testing.assert_allclose(d.loss, G[f"{name}/{kind}/loss"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.logits_gradient(d, inp["logits"]), G[f"{name}/{kind}/grad_logits"], atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_golden(name, kind):
    import torch
    import tf_seq2seq_losses_amd as ctc
    from tf_seq2seq_losses_amd import ops, _lib
    inp = _inp(name)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    x = t["logits"].clone().requires_grad_(True)
    fn = ctc.classic_ctc_loss if kind == "classic" else ctc.simple_ctc_loss
    loss = fn(t["labels"], x, t["label_length"], t["logit_length"], 0)
    (g,) = torch.autograd.grad(loss.sum(), x)
    want = G[f"{name}/{kind}/loss"]
    fin = np.isfinite(want)
    got = loss.detach().cpu().numpy()
    assert np.array_equal(np.isfinite(got), fin)
    assert (np.abs(got[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))).max() < TOL
    assert np.abs(g.cpu().numpy() - G[f"{name}/{kind}/grad_logits"]).max() < TOL
    if f"{name}/{kind}/hessian_logits" in G.files:
        p = ops.Prepared(t["labels"], t["logits"], t["label_length"], t["logit_length"], 0)
        for wrt, key in ((_lib.WRT_LOGITS, "hessian_logits"),):
            _, _, h = ops.hessian(ops.KINDS[kind], wrt, p)
            assert np.abs(h.cpu().numpy() - G[f"{name}/{kind}/{key}"]).max() < TOL
        lp = torch.log_softmax(t["logits"], 2)
        p = ops.Prepared(t["labels"], lp, t["label_length"], t["logit_length"], 0)
        _, _, h = ops.hessian(ops.KINDS[kind], _lib.WRT_LOGPROBS, p)
        assert np.abs(h.cpu().numpy() - G[f"{name}/{kind}/hessian_logprobs"]).max() < TOL
