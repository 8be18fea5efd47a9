"""Pins the oracle (oracle/ctc_oracle.py) against every known-answer value held by the reference's own
unit tests (tests/golden/reference_known_answers.json).  CPU only."""
import numpy as np
import pytest

from oracle import ctc_oracle as O
from tests._cases import load_known_answers, case_inputs, check_case, assert_close, _num

KA = load_known_answers()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", KA["cases"], ids=[c["id"] for c in KA["cases"]])
def test_known_answer(case, dtype):
    inp = case_inputs(case)
    data = O.LOSS_DATA[case["kind"]](inp["labels"], inp["logprobas"], inp["label_length"], inp["logit_length"],
                                     inp["blank"], dtype=dtype)
    check_case(case, data)
    if case.get("gamma00_equals_alpha"):
        # tests/test_hessian.py:62-87
        assert np.array_equal(np.exp(data.alpha), np.exp(data.gamma[:, 0, 0]))


def test_tools_logsumexp():
    t = KA["tools"]["logsumexp"]
    x = np.asarray(_num(t["x"]), dtype=np.float32)
    y = np.asarray(_num(t["y"]), dtype=np.float32)
    assert_close(O.logsumexp(x, y), t["expected"], t["places"], "logsumexp")


def test_tools_segment_logsumexp():
    t = KA["tools"]["unsorted_segment_logsumexp"]
    out = O.unsorted_segment_logsumexp(np.asarray(_num(t["data"]), dtype=np.float32),
                                       np.asarray(t["segment_ids"]), t["num_segments"])
    assert_close(out, t["expected"], t["places"], "segment lse")


@pytest.mark.parametrize("case", KA["shape_cases"], ids=[c["id"] for c in KA["shape_cases"]])
def test_shape_cases(case):
    B, T, V = case["logits_shape"]
    logits = np.full((B, T, V), case.get("logits_fill", 0.0), dtype=np.float32)
    if "labels" in case:
        labels = np.asarray(case["labels"], dtype=np.int32)
        label_length = np.asarray(case["label_length"], dtype=np.int32)
        logit_length = np.asarray(case["logit_length"], dtype=np.int32)
    else:
        labels = np.zeros(case["labels_shape"], dtype=np.int32)
        label_length = np.zeros((B,), dtype=np.int32)
        logit_length = np.zeros((B,), dtype=np.int32)
    data = O.ctc_loss(case["kind"], labels, logits, label_length, logit_length, case.get("blank", 0))
    assert list(data.loss.shape) == case.get("loss_shape", [B])
    g = O.logits_gradient(data, logits)
    assert list(g.shape) == case.get("grad_shape", [B, T, V])
    if "mean_loss" in case:
        assert np.mean(data.loss) == np.inf
    if "hessian_shape" in case:
        assert list(O.logits_hessian(data, logits).shape) == case["hessian_shape"]
        # README sample 0: labels 1,2,2,1 in 5 frames has the single path class 1,2,_,2,1
        assert abs(data.loss[0] - 5 * np.log(3.0)) < 1e-12
