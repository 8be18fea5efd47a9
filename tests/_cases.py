"""Shared helpers for tests: loads the golden known-answer table and turns records into arrays."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _num(x):
    if isinstance(x, str):
        return float(x)
    if isinstance(x, list):
        return [_num(v) for v in x]
    return x


def load_known_answers():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


def case_inputs(case):
    """-> dict(labels, logits, logprobas, label_length, logit_length, blank) as numpy (float32 inputs)."""
    if "P" in case:
        with np.errstate(divide="ignore"):
            logits = np.log(np.asarray(case["P"], dtype=np.float32))
    else:
        logits = np.asarray(_num(case["logits"]), dtype=np.float32)
    m = logits.max(axis=2, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        logprobas = (logits - (m + np.log(np.exp(logits - m).sum(axis=2, keepdims=True)))).astype(np.float32)
    return dict(
        labels=np.asarray(case["labels"], dtype=np.int32),
        logits=logits,
        logprobas=logprobas,
        label_length=np.asarray(case["label_length"], dtype=np.int32),
        logit_length=np.asarray(case["logit_length"], dtype=np.int32),
        blank=int(case.get("blank", 0)),
    )


def assert_close(actual, expected, places, what=""):
    """The reference's assert_tensors_almost_equal (tests/test_ctc_losses.py:28-47): L-inf norm,
    places=None means exact equality.  inf == inf counts as equal."""
    a = np.asarray(actual, dtype=np.float64)
    # the reference compares float32 tensors: python-float expectations are rounded to float32 first
    e = np.asarray(_num(expected), dtype=np.float32).astype(np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    same_inf = np.isinf(a) & np.isinf(e) & (np.sign(a) == np.sign(e))
    with np.errstate(invalid="ignore"):
        diff = np.where(same_inf, 0.0, np.abs(a - e))
    err = float(diff.max()) if diff.size else 0.0
    tol = 0.0 if places is None else 0.5 * 10.0 ** (-places)
    assert err <= tol and not np.isnan(err), f"{what}: max|diff|={err} > {tol}\nactual={a}\nexpected={e}"


def check_case(case, data, exact_ulps=0):
    """Checks one known-answer record against a loss-data-like object exposing
    alpha, beta, loss, logarithmic_logproba_gradient, gradient, hessian (numpy arrays).
    exact_ulps: slack (in float32 ulps of the expected value) granted where the reference asserts
    exact equality on a non-trivial float (used for the HIP path, documented in the test)."""
    def tol_places(key):
        return case.get(key)

    def close(actual, expected, places, what):
        if places is None and exact_ulps:
            a = np.asarray(actual, dtype=np.float64)
            e = np.asarray(_num(expected), dtype=np.float32).astype(np.float64)
            assert a.shape == e.shape
            same_inf = np.isinf(a) & np.isinf(e)
            ulp = np.spacing(np.abs(np.where(np.isinf(e), 0, e)).astype(np.float32)).astype(np.float64)
            with np.errstate(invalid="ignore"):
                bad = ~same_inf & ~(np.abs(a - e) <= exact_ulps * ulp)
            assert not bad.any(), f"{case['id']} {what}: {a} vs {e}"
        else:
            assert_close(actual, expected, places, f"{case['id']} {what}")

    if "exp_alpha" in case:
        close(np.exp(data.alpha), case["exp_alpha"], tol_places("exp_alpha_places"), "exp(alpha)")
    if "exp_beta" in case:
        close(np.exp(data.beta), case["exp_beta"], tol_places("exp_beta_places"), "exp(beta)")
    if "loss" in case:
        close(data.loss, case["loss"], tol_places("loss_places"), "loss")
    if "loss_less_than" in case:
        assert float(np.asarray(data.loss)[0]) < case["loss_less_than"], f"{case['id']} loss"
    if "exp_lg" in case:
        close(np.exp(data.logarithmic_logproba_gradient), case["exp_lg"], tol_places("exp_lg_places"), "exp(lg)")
    if "gradient" in case:
        close(data.gradient, case["gradient"], tol_places("gradient_places"), "gradient")
    if "hessian_zero_shape" in case:
        close(data.hessian, np.zeros(case["hessian_zero_shape"]), tol_places("hessian_places"), "hessian")
