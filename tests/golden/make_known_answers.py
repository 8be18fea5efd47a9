"""Writes tests/golden/reference_known_answers.json.

The TensorFlow reference cannot be imported in the build container (no tensorflow), so the
only reference-held values that exist for this path are the hand-computed known answers inside
the reference's own unit tests.  This script is the transcription of that DATA (inputs and
expected outputs only -- no reference code), one record per assertion group, each citing the
test it comes from (paths relative to /root/reference).

"P" means the test builds log-probabilities as log_softmax(log(P)); "logits" means
log_softmax(logits).  `places` is the reference's assertAlmostEqual precision
(max|a-b| < 0.5e-places, tests/test_ctc_losses.py:28-47); null = exact equality.
Run:  python tests/golden/make_known_answers.py
"""
import json
import math
import os

INF = "inf"

CASES = [
    dict(id="classic_single_logit", source="tests/test_classic_ctc_loss.py:33-65", kind="classic",
         P=[[[0, 1, 0]]], labels=[[1]], label_length=[1], logit_length=[1], blank=0,
         exp_alpha=[[[[1, 0], [0, 0]], [[0, 0], [0, 1]]]], exp_alpha_places=None,
         exp_beta=[[[[1, 1], [0, 1]], [[0, 0], [1, 1]]]], exp_beta_places=None,
         loss=[0.0], loss_places=None,
         exp_lg=[[[0.0, 1.0, 0.0]]], exp_lg_places=6),
    dict(id="classic_closed_state", source="tests/test_classic_ctc_loss.py:67-105", kind="classic",
         P=[[[0, 1, 0], [1, 0, 0]]], labels=[[1]], label_length=[1], logit_length=[2], blank=0,
         exp_alpha=[[[[1, 0], [0, 0]], [[0, 0], [0, 1]], [[0, 0], [1, 0]]]], exp_alpha_places=None,
         exp_beta=[[[[1, 1], [0, 1]], [[0, 0], [1, 1]], [[0, 0], [1, 1]]]], exp_beta_places=None,
         loss=[0.0], loss_places=None,
         exp_lg=[[[0.0, 1.0, 0.0], [1.0, 0.0, 0.0]]], exp_lg_places=6),
    dict(id="classic_simple_case", source="tests/test_classic_ctc_loss.py:107-144", kind="classic",
         P=[[[0, 1, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1], [0, 1, 0]]], labels=[[1, 2, 2, 1]],
         label_length=[4], logit_length=[5], blank=0,
         loss_less_than=1e-6,
         exp_lg=[[[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0]]],
         exp_lg_places=6),
    dict(id="classic_length_two", source="tests/test_classic_ctc_loss.py:169-199", kind="classic",
         logits=[[[0, 0, 0], [0, 0, 0]], [[0, 0, 0], [0, 0, 0]]], labels=[[1, 2], [1, 2]],
         label_length=[2, 1], logit_length=[2, 2], blank=0,
         loss=[math.log(9.0), math.log(3.0)], loss_places=6,
         gradient=[[[0.0, -1.0, 0.0], [0.0, 0.0, -1.0]],
                   [[-1 / 3, -2 / 3, 0.0], [-1 / 3, -2 / 3, 0.0]]], gradient_places=6),
    dict(id="classic_too_short_logit", source="tests/test_classic_ctc_loss.py:201-241", kind="classic",
         logits=[[[0, 0, 0], [0, 0, 0]]], labels=[[1, 1]], label_length=[2], logit_length=[2], blank=0,
         loss=[INF], loss_places=None,
         gradient=[[[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]], gradient_places=None,
         hessian_zero_shape=[1, 2, 3, 2, 3], hessian_places=None),
    dict(id="classic_repeated_token", source="tests/test_classic_ctc_loss.py:243-262", kind="classic",
         logits=[[[0, 0, 0]] * 3], labels=[[1, 1]], label_length=[2], logit_length=[3], blank=0,
         loss=[math.log(27.0)], loss_places=6),
    dict(id="classic_single_token", source="tests/test_classic_ctc_loss.py:264-283", kind="classic",
         logits=[[[0, 0, 0]] * 3], labels=[[1]], label_length=[1], logit_length=[3], blank=0,
         loss=[math.log(27.0 / 6.0)], loss_places=6),
    dict(id="classic_wrong_prediction", source="tests/test_classic_ctc_loss.py:285-307", kind="classic",
         logits=[[[0, 0, 100]]], labels=[[1]], label_length=[1], logit_length=[1], blank=0,
         loss=[100.0], loss_places=None,
         gradient=[[[0.0, -1.0, 0.0]]], gradient_places=None),
    dict(id="simplified_simple_case", source="tests/test_simplified_ctc_loss.py:35-91", kind="simplified",
         P=[[[0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]]], labels=[[1, 2, 1]],
         label_length=[3], logit_length=[5], blank=0,
         exp_alpha=[[[1, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]]],
         exp_alpha_places=None,
         exp_beta=[[[1, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]]],
         exp_beta_places=None,
         loss_less_than=1e-6,
         gamma00_equals_alpha=True),  # tests/test_hessian.py:62-87
    dict(id="simplified_non_zero_blank", source="tests/test_simplified_ctc_loss.py:93-115", kind="simplified",
         P=[[[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0], [1, 0, 0]]], labels=[[0, 2, 0]],
         label_length=[3], logit_length=[5], blank=1,
         loss_less_than=1e-6),
    dict(id="simplified_shorter_lengths", source="tests/test_simplified_ctc_loss.py:117-138", kind="simplified",
         P=[[[1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0]]], labels=[[1, 0]],
         label_length=[1], logit_length=[3], blank=0,
         loss=[0.0], loss_places=None),
    dict(id="simplified_label_longer_than_logit", source="tests/test_simplified_ctc_loss.py:140-160",
         kind="simplified",
         logits=[[[0, 0, 0]]], labels=[[1, 2]], label_length=[2], logit_length=[1], blank=0,
         loss=[INF], loss_places=None,
         gradient=[[[0.0, 0.0, 0.0]]], gradient_places=None),
    dict(id="simplified_large_loss", source="tests/test_simplified_ctc_loss.py:162-183", kind="simplified",
         logits=[[[1e10, 0.0, 0.0]]], labels=[[1]], label_length=[1], logit_length=[1], blank=0,
         loss=[1e10], loss_places=None,
         gradient=[[[0.0, -1.0, 0.0]]], gradient_places=None),
    dict(id="simplified_length_one", source="tests/test_simplified_ctc_loss.py:208-230", kind="simplified",
         logits=[[[0, 0, 0]]], labels=[[1]], label_length=[1], logit_length=[1], blank=0,
         loss=[math.log(3.0)], loss_places=8,
         gradient=[[[0.0, -1.0, 0.0]]], gradient_places=6),
    dict(id="simplified_length_two", source="tests/test_simplified_ctc_loss.py:232-258", kind="simplified",
         logits=[[[0, 0, 0], [0, 0, 0]]], labels=[[1, 2]], label_length=[2], logit_length=[2], blank=0,
         loss=[2 * math.log(3.0)], loss_places=8,
         gradient=[[[0.0, -1.0, 0.0], [0.0, 0.0, -1.0]]], gradient_places=6),
    dict(id="simplified_hessian_single_logit", source="tests/test_hessian.py:37-60", kind="simplified",
         P=[[[1 / 3, 1 / 3, 1 / 3]]], labels=[[1]], label_length=[1], logit_length=[1], blank=0,
         gradient=[[[0.0, -1.0, 0.0]]], gradient_places=6,
         hessian_zero_shape=[1, 1, 3, 1, 3], hessian_places=6),
]

# Plumbing cases of the public functions (shape-only assertions in the reference).
SHAPE_CASES = [
    dict(id="classic_zero_batch", source="tests/test_classic_ctc_loss.py:309-330", kind="classic",
         logits_shape=[0, 4, 3], labels_shape=[0, 2], loss_shape=[0], grad_shape=[0, 4, 3]),
    dict(id="simplified_zero_batch", source="tests/test_simplified_ctc_loss.py:345-366", kind="simplified",
         logits_shape=[0, 4, 3], labels_shape=[0, 2], loss_shape=[0], grad_shape=[0, 4, 3]),
    dict(id="simplified_zero_logit_length", source="tests/test_simplified_ctc_loss.py:322-343",
         kind="simplified", logits_shape=[1, 0, 3], labels=[[1, 2]], label_length=[2], logit_length=[2],
         mean_loss=INF, grad_shape=[1, 0, 3]),
    dict(id="readme_example", source="README.md:50-71; tests/test_hessian.py:185-213", kind="classic",
         logits_shape=[2, 5, 3], logits_fill=0.0, labels=[[1, 2, 2, 1], [1, 2, 1, 0]],
         label_length=[4, 3], logit_length=[5, 4], blank=0, hessian_shape=[2, 5, 3, 5, 3]),
]

# tools.py primitives (tests/test_tools.py:37-51 and :137-148)
TOOLS = dict(
    logsumexp=dict(source="tests/test_tools.py:37-51",
                   x=[-3.0753517, "-inf", "-inf"], y=[-1e12, -0.4283799, "-inf"],
                   expected=[-3.0753517, -0.4283799, "-inf"], places=6),
    unsorted_segment_logsumexp=dict(source="tests/test_tools.py:137-148",
                                    data=[0.0, "-inf", 0.0, "-inf"], segment_ids=[0, 1, 0, 1], num_segments=2,
                                    expected=[math.log(2.0), "-inf"], places=6),
)

if __name__ == "__main__":
    out = dict(
        note="Known-answer values transcribed from the unit tests of alexeytochin/tf_seq2seq_losses v0.3.0 "
             "(data only; see make_known_answers.py).  Not generated by running the reference: TensorFlow "
             "is not installable here.",
        cases=CASES, shape_cases=SHAPE_CASES, tools=TOOLS)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(CASES), "cases")
