"""Pins the C restatement (oracle/ctc_oracle.c) against the NumPy oracle (itself pinned against the reference's
known answers) and against the known-answer table directly.  CPU only."""
import numpy as np
import pytest

from oracle import ctc_oracle as O
from oracle import c_oracle as C
from tests._cases import load_known_answers, case_inputs

KA = load_known_answers()


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("B,T,V,seed", [(8, 20, 8, 0), (8, 64, 10, 1), (5, 33, 7, 4)])
def test_c_oracle_matches_numpy_oracle(kind, B, T, V, seed):
    inp = O.generate_ctc_loss_inputs(B, T, seed, V)
    ref = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    loss, grad = C.loss_grad(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    fin = np.isfinite(ref.loss)
    assert np.array_equal(np.isfinite(loss), fin)
    assert np.abs(loss[fin] - ref.loss[fin]).max() < 1e-9
    assert np.abs(grad - O.logits_gradient(ref, inp["logits"])).max() < 1e-9


@pytest.mark.parametrize("case", [c for c in KA["cases"] if "logits" in c and "loss" in c],
                         ids=[c["id"] for c in KA["cases"] if "logits" in c and "loss" in c])
def test_c_oracle_known_answer_losses(case):
    inp = case_inputs(case)
    loss, _ = C.loss_grad(case["kind"], inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], inp["blank"])
    exp = np.array([float(v) for v in case["loss"]])
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(loss), fin)
    if fin.any():
        assert (np.abs(loss[fin] - exp[fin]) / np.maximum(1, np.abs(exp[fin]))).max() < 1e-6


def test_c_oracle_float32_build_and_threads():
    inp = O.generate_ctc_loss_inputs(6, 40, 3, 9)
    l64, g64 = C.loss_grad("classic", inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    l32, g32 = C.loss_grad("classic", inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0,
                           precision="f32", n_threads=2)
    fin = np.isfinite(l64)
    assert np.abs(l32[fin] - l64[fin]).max() < 1e-3 and np.abs(g32 - g64).max() < 1e-4
    assert C.num_threads() >= 1
