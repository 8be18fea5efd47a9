"""CPU oracle for the CTC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy restatement of the algorithm of alexeytochin/tf_seq2seq_losses
(reference checked out read-only at /root/reference, v0.3.0).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The shipped package ``tf_seq2seq_losses_amd`` never imports anything under ``oracle/``.

Parity status: PINNED at known-answer level.  The TensorFlow reference cannot be
imported in the build container (``ModuleNotFoundError: tensorflow`` -- an ordinary
Python error, nothing was denied), so no reference-generated vectors exist.  The oracle
is pinned by every hand-computable known-answer value the reference's own unit tests
hold (``tests/golden/reference_known_answers.json``, transcribed from
``tests/test_classic_ctc_loss.py``, ``tests/test_simplified_ctc_loss.py``,
``tests/test_hessian.py``, ``tests/test_tools.py``), and cross-checked against three
independent sources that are not the reference: brute-force path enumeration, torch's
CPU ``ctc_loss`` and finite differences.  Random-input values that the reference tests
draw with ``tf.random`` are unpinned (RNG not reproducible without TF).

Every function cites the reference lines it restates.  Arithmetic is done in the dtype
given (float64 = arbiter, float32 = the reference's own precision).  The structure follows
the reference on purpose (dense [B, T, L(, 2)] tables, one python loop over T, log space,
token scatter through a segmented log-sum-exp) so that it can be read side by side with it.

Index conventions (identical to the reference):
    B batch, T = logits.shape[1], V tokens, U = max(label_length), L = U + 1,
    alpha/beta: classic [B, T+1, L, 2] (s=0 closed, s=1 open), simplified [B, T+1, L].
"""
from __future__ import annotations

import itertools
from functools import cached_property

import numpy as np

NEG_INF = -np.inf


# ----------------------------------------------------------------------------------------------
# tools.py
# ----------------------------------------------------------------------------------------------
def reduce_logsumexp(x: np.ndarray, axis, keepdims: bool = False) -> np.ndarray:
    """tf.reduce_logsumexp semantics (used at tools.py:37 and many call sites):
    max-shifted, with a non-finite max replaced by 0 so that all -inf gives -inf, not NaN."""
    x = np.asarray(x)
    if x.size == 0 or any(x.shape[a] == 0 for a in np.atleast_1d(axis)):
        shape = list(x.shape)
        for a in sorted(np.atleast_1d(axis) % x.ndim, reverse=True):
            if keepdims:
                shape[a] = 1
            else:
                del shape[a]
        return np.full(shape, NEG_INF, dtype=x.dtype)
    raw_max = np.max(x, axis=axis, keepdims=True)
    my_max = np.where(np.isfinite(raw_max), raw_max, 0).astype(x.dtype)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        s = np.sum(np.exp(x - my_max), axis=axis, keepdims=True)
        out = np.log(s) + my_max
    if not keepdims:
        out = np.squeeze(out, axis=tuple(np.atleast_1d(axis)))
    return out.astype(x.dtype)


def logit_to_logproba(logit: np.ndarray, axis: int = 2) -> np.ndarray:
    """tools.py:27-40 : x - logsumexp(x, axis, keepdims)."""
    with np.errstate(invalid="ignore"):
        return logit - reduce_logsumexp(logit, axis=axis, keepdims=True)


def apply_logarithmic_mask(tensor: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """tools.py:43-54 : tensor + log(float(mask))  (0 or -inf)."""
    with np.errstate(divide="ignore"):
        return tensor + np.log(mask.astype(tensor.dtype))


def _softplus(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore", invalid="ignore"):
        return np.where(x > 30, x, np.log1p(np.exp(np.minimum(x, 30))))


def logsumexp(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """tools.py:57-71 : two-argument stable log(e^x + e^y); x == y branch returns x + log 2, so
    (-inf, -inf) -> -inf without NaN."""
    x, y = np.broadcast_arrays(np.asarray(x), np.asarray(y))
    dt = np.result_type(x, y)
    with np.errstate(invalid="ignore", over="ignore"):
        lo = y + _softplus(x - y)
        hi = x + _softplus(y - x)
        eq = x + dt.type(np.log(2.0))
    return np.where(x < y, lo, np.where(x > y, hi, eq)).astype(dt)


def unsorted_segment_logsumexp(data: np.ndarray, segment_ids: np.ndarray, num_segments: int) -> np.ndarray:
    """tools.py:95-119 : segment max, exp(data - max[ids]), segment sum, log, + max.
    Empty segments come out as -inf (tf.math.unsorted_segment_max gives the lowest float
    there, so the reference produces lowest + log(0) = -inf as well)."""
    data = np.asarray(data)
    rest = data.shape[1:]
    data_max = np.full((num_segments,) + rest, NEG_INF, dtype=data.dtype)
    np.maximum.at(data_max, segment_ids, data)
    safe_max = np.where(np.isfinite(data_max), data_max, 0).astype(data.dtype)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        normed = np.exp(data - safe_max[segment_ids])
        sums = np.zeros((num_segments,) + rest, dtype=data.dtype)
        np.add.at(sums, segment_ids, normed)
        out = safe_max + np.log(sums)
    return out.astype(data.dtype)


def expand_many_dims(x: np.ndarray, axes) -> np.ndarray:
    """tools.py:294-312."""
    for a in axes:
        x = np.expand_dims(x, a)
    return x


# ----------------------------------------------------------------------------------------------
# base_loss.py : BaseCtcLossData
# ----------------------------------------------------------------------------------------------
class BaseCtcLossData:
    """Restates base_loss.py:102-543.  ``logprobas`` are log-probabilities treated as independent
    variables (base_loss.py:71-99), not necessarily normalised."""

    n_state_dims = 0  # trailing state dims of alpha rows: 0 simplified, 1 classic

    def __init__(self, labels, logprobas, label_length, logit_length, blank_index=0, dtype=np.float64):
        logprobas = np.asarray(logprobas)
        labels = np.asarray(labels)
        label_length = np.asarray(label_length)
        logit_length = np.asarray(logit_length)
        # base_loss.py:129-138
        assert logprobas.ndim == 3
        assert labels.ndim == 2
        assert logit_length.ndim == 1
        assert label_length.ndim == 1
        assert logprobas.shape[0] == labels.shape[0] == logit_length.shape[0] == label_length.shape[0]
        self.dtype = np.dtype(dtype)
        self._logprobas = logprobas.astype(self.dtype)
        self._original_label = labels.astype(np.int64)
        self._logit_length = logit_length.astype(np.int64)
        self._label_length = label_length.astype(np.int64)
        self._blank = int(blank_index)

    # -- shapes (base_loss.py:470-534) ---------------------------------------------------------
    @property
    def B(self):
        return self._logprobas.shape[0]

    @property
    def T(self):
        return self._logprobas.shape[1]

    @property
    def V(self):
        return self._logprobas.shape[2]

    @cached_property
    def U(self):
        """base_loss.py:482-486 : max(label_length), 0 when the batch is empty."""
        return int(self._label_length.max()) if self.B > 0 else 0

    @property
    def L(self):
        return self.U + 1

    @cached_property
    def _logit_length_mask(self):
        """base_loss.py:500-506 : [B, T] bool."""
        return np.arange(self.T)[None, :] < self._logit_length[:, None]

    @cached_property
    def _label_length_mask(self):
        """base_loss.py:508-513 : [B, L] bool."""
        return np.arange(self.L)[None, :] < self._label_length[:, None]

    @cached_property
    def _label(self):
        """base_loss.py:395-418 : truncate / right-pad to L columns, overwrite l >= label_length by blank."""
        lab = self._original_label
        if lab.shape[1] > self.U:
            lab = lab[:, : self.L]
        if lab.shape[1] < self.L:
            pad = np.full((self.B, self.L - lab.shape[1]), self._blank, dtype=np.int64)
            lab = np.concatenate([lab, pad], axis=1)
        mask = np.arange(self.L)[None, :] < self._label_length[:, None]
        return np.where(mask, lab, self._blank)

    @cached_property
    def _preceded_label(self):
        """base_loss.py:519-525 : cyclic roll by +1 along l."""
        return np.roll(self._label, 1, axis=1)

    @cached_property
    def _logproba(self):
        """base_loss.py:378-393 : frames t >= logit_length become log one_hot(blank)."""
        blank_lp = np.full((1, 1, self.V), NEG_INF, dtype=self.dtype)
        blank_lp[0, 0, self._blank] = 0.0
        return np.where(self._logit_length_mask[:, :, None], self._logprobas, blank_lp)

    @cached_property
    def _blank_logproba(self):
        """base_loss.py:365-371 : [B, T]."""
        return self._logproba[:, :, self._blank]

    def _gather_tokens(self, params, indices):
        """tf.gather(params[B,T,V], indices[B,L], axis=2, batch_dims=1) -> [B,T,L]."""
        return np.take_along_axis(params, indices[:, None, :].repeat(params.shape[1], axis=1), axis=2) \
            if params.shape[1] > 0 else np.zeros((self.B, 0, indices.shape[1]), dtype=params.dtype)

    @cached_property
    def _expected_token_logproba(self):
        """base_loss.py:328-344 : y[b,t,l] = lp[b,t,label[b,l]] + log(l < label_length[b])."""
        y = self._gather_tokens(self._logproba, self._label)
        return apply_logarithmic_mask(y, self._label_length_mask[:, None, :])

    def _select_from_act(self, act: np.ndarray, label: np.ndarray) -> np.ndarray:
        """base_loss.py:420-468 : out[b,a,t,k,...] = LSE_{l: label[b,l]=k} act[b,a,t,l,...].
        act: [B, A, T, L, ...] -> [B, A, T, V, ...]."""
        data = np.swapaxes(act, 1, 3)  # [B, L, T, A, ...]
        shp = data.shape
        data = data.reshape((shp[0] * shp[1],) + shp[2:])
        segment_ids = (label + np.arange(self.B)[:, None] * self.V).reshape(-1)
        out = unsorted_segment_logsumexp(data, segment_ids, self.B * self.V)
        out = out.reshape((self.B, self.V) + shp[2:])  # [B, V, T, A, ...]
        return np.swapaxes(out, 1, 3)  # [B, A, T, V, ...]

    # -- derived quantities (base_loss.py:186-298) -----------------------------------------------
    @cached_property
    def logarithmic_logproba_gradient(self):
        """base_loss.py:270-298."""
        if self.T == 0 or self.B == 0:
            return np.zeros((self.B, self.T, self.V), dtype=self.dtype)
        with np.errstate(invalid="ignore"):
            lg = self.loss.reshape(-1, 1, 1) + self._combine_transition_probabilities(
                a=self.alpha[:, :-1], b=self.beta[:, 1:])
        lg = np.where((self.loss == np.inf)[:, None, None], NEG_INF, lg)
        return apply_logarithmic_mask(lg, self._logit_length_mask[:, :, None]).astype(self.dtype)

    @cached_property
    def gradient(self):
        """base_loss.py:262-268 : -exp(lg)."""
        return -np.exp(self.logarithmic_logproba_gradient)

    @cached_property
    def hessian(self):
        """base_loss.py:186-260 : via the all-pairs transition tensor gamma (O(T^2 L^2) memory)."""
        B, T, V = self.B, self.T, self.V
        if T == 0 or B == 0:
            return np.zeros((B, T, V, T, V), dtype=self.dtype)
        alpha_gamma = self._combine_transition_probabilities(a=self.alpha[:, :-1], b=self.gamma[:, 1:])
        # [B, T, V, T+1, L(,2)]
        agb = self._combine_transition_probabilities(a=alpha_gamma[:, :, :, :-1], b=self.beta[:, 1:])
        # [B, T, V, T, V]
        with np.errstate(invalid="ignore"):
            agb_loss = expand_many_dims(self.loss, [1, 2, 3, 4]) + agb
        n = T * V
        first = agb_loss.reshape(B, n, n).copy()
        diag = self.logarithmic_logproba_gradient.reshape(B, n)
        idx = np.arange(n)
        first[:, idx, idx] = diag  # tf.linalg.set_diag, base_loss.py:205-221
        first = first.reshape(B, T, V, T, V)
        mask = np.triu(np.ones((T, T), dtype=bool))[None, :, None, :, None]  # band_part(0,-1): t1 <= t2
        sym = np.where(mask, first, np.transpose(first, (0, 3, 4, 1, 2)))
        with np.errstate(invalid="ignore"):
            hess = -np.exp(sym) + expand_many_dims(self.gradient, [3, 4]) * expand_many_dims(self.gradient, [1, 2])
        hess = np.where(expand_many_dims(self.loss == np.inf, [1, 2, 3, 4]), 0.0, hess)
        hess = np.where(expand_many_dims(self._logit_length_mask, [2, 3, 4]), hess, 0.0)
        hess = np.where(expand_many_dims(self._logit_length_mask, [1, 2, 4]), hess, 0.0)
        return hess.astype(self.dtype)

    # to be provided by the lattice variants
    alpha: np.ndarray
    beta: np.ndarray
    gamma: np.ndarray
    loss: np.ndarray

    def _combine_transition_probabilities(self, a, b):
        raise NotImplementedError

    # tools.py:191-277 (unfold): T sequential steps, slices stacked on a new leading time axis.
    @staticmethod
    def _unfold(init, step, d_i, num_iters):
        slices = [None] * (num_iters + 1)
        if d_i == 1:
            slices[0] = init
            for i in range(num_iters):
                slices[i + 1] = step(slices[i], i)
        else:
            slices[num_iters] = init
            for i in range(num_iters - 1, -1, -1):
                slices[i] = step(slices[i + 1], i)
        return np.stack(slices, axis=0)


# ----------------------------------------------------------------------------------------------
# classic_ctc_loss.py : ClassicCtcLossData
# ----------------------------------------------------------------------------------------------
class ClassicCtcLossData(BaseCtcLossData):
    n_state_dims = 1

    # transition tables, classic_ctc_loss.py:464-563
    @cached_property
    def _closed_to_open_diagonal(self):
        return self._expected_token_logproba  # :494-501

    @cached_property
    def _open_to_open_diagonal(self):
        """:478-492 : y masked where label[l] == label[l-1] (cyclic)."""
        no_repeat = self._label != np.roll(self._label, 1, axis=1)
        return apply_logarithmic_mask(self._closed_to_open_diagonal, no_repeat[:, None, :])

    @cached_property
    def _any_to_open_diagonal(self):
        """:464-476 : [B, T, L, state]."""
        return np.stack([self._closed_to_open_diagonal, self._open_to_open_diagonal], axis=3)

    @cached_property
    def _not_blank_horizontal(self):
        """:528-543 : lp with the blank column masked, gathered at prev[l]."""
        mask = np.ones((1, 1, self.V), dtype=bool)
        mask[0, 0, self._blank] = False
        nb = apply_logarithmic_mask(self._logproba, mask)
        return self._gather_tokens(nb, np.roll(self._label, 1, axis=1))

    @cached_property
    def _previous_label_token(self):
        """:545-558 : lp gathered at prev[l], unmasked."""
        return self._gather_tokens(self._logproba, self._preceded_label)

    @cached_property
    def _horizontal(self):
        """:503-526 : H[b,t,l,next,prev] = [[bl, bl], [-inf, rep]]."""
        bl = np.broadcast_to(self._blank_logproba[:, :, None, None], (self.B, self.T, self.L, 2))
        nb = np.stack([np.full_like(self._not_blank_horizontal, NEG_INF), self._not_blank_horizontal], axis=3)
        return np.stack([bl, nb], axis=3).astype(self.dtype)

    # alpha, :379-462
    @cached_property
    def alpha(self):
        a0 = np.full((self.B, self.L, 2), NEG_INF, dtype=self.dtype)
        if self.B:
            a0[:, 0, 0] = 0.0
        H, D = self._horizontal, self._any_to_open_diagonal

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                horizontal = reduce_logsumexp(H[:, i] + prev[:, :, None, :], axis=3)
                diag = reduce_logsumexp(D[:, i] + prev, axis=2)
            moved = np.roll(diag, 1, axis=1)
            diagonal = np.stack([np.full_like(moved, NEG_INF), moved], axis=2)
            return logsumexp(horizontal, diagonal)

        out = self._unfold(a0, step, 1, self.T)  # [T+1, B, L, 2]
        return np.transpose(out, (1, 0, 2, 3))

    # beta, :310-377
    @cached_property
    def beta(self):
        bT = np.full((self.B, self.L), NEG_INF, dtype=self.dtype)
        if self.B:
            bT[np.arange(self.B), np.minimum(self._label_length, self.L - 1)] = 0.0
        bT = np.repeat(bT[:, :, None], 2, axis=2)
        H, D = self._horizontal, self._any_to_open_diagonal

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                horizontal = reduce_logsumexp(H[:, i] + prev[:, :, :, None], axis=2)
                diagonal = D[:, i] + np.roll(prev[:, :, 1:], -1, axis=1)
            return logsumexp(horizontal, diagonal)

        out = self._unfold(bT, step, -1, self.T)
        return np.transpose(out, (1, 0, 2, 3))

    # loss, :152-165
    @cached_property
    def loss(self):
        params = reduce_logsumexp(self.alpha[:, -1], axis=-1)  # [B, L]
        if self.B == 0:
            return np.zeros((0,), dtype=self.dtype)
        return -params[np.arange(self.B), self._label_length]

    # gamma, :167-308
    @cached_property
    def gamma(self):
        B, T, L = self.B, self.T, self.L
        eye = np.eye(2 * L, dtype=self.dtype).reshape(1, 1, L, 2, L, 2)
        with np.errstate(divide="ignore"):
            diagonal_gamma = np.broadcast_to(np.log(eye), (B, T + 1, L, 2, L, 2)).copy()
        H, D = self._horizontal, self._any_to_open_diagonal

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                hs = expand_many_dims(H[:, i], [1, 2, 3]) + np.expand_dims(prev, 5)
                horizontal = reduce_logsumexp(hs, axis=6)
                diag = reduce_logsumexp(expand_many_dims(D[:, i], [1, 2, 3]) + prev, axis=5)
            moved = np.roll(diag, 1, axis=4)
            diagonal = np.stack([np.full_like(moved, NEG_INF), moved], axis=5)
            new = logsumexp(horizontal, diagonal)
            cond = (np.arange(T + 1) <= i).reshape(1, -1, 1, 1, 1, 1)
            return np.where(cond, new, diagonal_gamma)

        fwd = self._unfold(diagonal_gamma, step, 1, T)  # [T+1(t2), B, T+1(t1), L, 2, L, 2]
        fwd = np.transpose(fwd, (1, 2, 3, 4, 0, 5, 6))
        mask = np.triu(np.ones((T + 1, T + 1), dtype=bool))
        return apply_logarithmic_mask(fwd, expand_many_dims(mask, [0, 2, 3, 5, 6]))

    # combine, :565-669
    def _combine_transition_probabilities(self, a, b):
        B, T, L = self.B, self.T, self.L
        dims_a = a.shape[1:-3]
        dims_b = b.shape[4:]
        a = a.reshape(B, -1, T, L, 2, 1)
        b = b.reshape(B, 1, T, L, 2, -1)
        with np.errstate(invalid="ignore"):
            ab = reduce_logsumexp(a, axis=4) + b[:, :, :, :, 0]
            horizontal_blank = expand_many_dims(self._blank_logproba, [1, 3]) + reduce_logsumexp(ab, axis=3)
            act = a[:, :, :, :, 1] + expand_many_dims(self._previous_label_token, [1, 4]) + b[:, :, :, :, 1]
            horizontal_non_blank = self._select_from_act(act, self._preceded_label)
            inp = a + expand_many_dims(self._any_to_open_diagonal, [1, 5]) + np.roll(b[:, :, :, :, 1:], -1, axis=3)
            act = reduce_logsumexp(inp, axis=4)
            diagonal_non_blank = self._select_from_act(act, self._label)
            non_blank = logsumexp(horizontal_non_blank, diagonal_non_blank)
        blank_mask = (np.arange(self.V) == self._blank).reshape(1, 1, 1, -1, 1)
        out = np.where(blank_mask, np.expand_dims(horizontal_blank, 3), non_blank)
        return out.reshape((B,) + dims_a + (T, self.V) + dims_b)


# ----------------------------------------------------------------------------------------------
# simplified_ctc_loss.py : SimplifiedCtcLossData
# ----------------------------------------------------------------------------------------------
class SimplifiedCtcLossData(BaseCtcLossData):
    n_state_dims = 0

    @cached_property
    def horizontal_step_log_proba(self):
        return self._blank_logproba  # :440-446

    @cached_property
    def diagonal_step_log_proba(self):
        return self._expected_token_logproba  # :448-454

    @cached_property
    def alpha(self):
        """:358-438."""
        a0 = np.full((self.B, self.L), NEG_INF, dtype=self.dtype)
        if self.B:
            a0[:, 0] = 0.0
        Hs, Ds = self.horizontal_step_log_proba, self.diagonal_step_log_proba

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                horizontal = Hs[:, i][:, None] + prev
                diagonal = Ds[:, i] + prev
            return logsumexp(horizontal, np.roll(diagonal, 1, axis=1))

        out = self._unfold(a0, step, 1, self.T)
        return np.transpose(out, (1, 0, 2))

    @cached_property
    def beta(self):
        """:291-356."""
        bT = np.full((self.B, self.L), NEG_INF, dtype=self.dtype)
        if self.B:
            bT[np.arange(self.B), np.minimum(self._label_length, self.L - 1)] = 0.0
        Hs, Ds = self.horizontal_step_log_proba, self.diagonal_step_log_proba

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                horizontal = Hs[:, i][:, None] + prev
                diagonal = Ds[:, i] + np.roll(prev, -1, axis=1)
            return logsumexp(horizontal, diagonal)

        out = self._unfold(bT, step, -1, self.T)
        return np.transpose(out, (1, 0, 2))

    @cached_property
    def loss(self):
        """:73-83."""
        if self.B == 0:
            return np.zeros((0,), dtype=self.dtype)
        return -self.alpha[:, -1][np.arange(self.B), self._label_length]

    @cached_property
    def gamma(self):
        """:85-191."""
        B, T, L = self.B, self.T, self.L
        with np.errstate(divide="ignore"):
            diagonal_gamma = np.log(np.eye(L, dtype=self.dtype)).reshape(1, 1, L, L)
        init = np.broadcast_to(diagonal_gamma, (B, T + 1, L, L)).copy()
        Hs, Ds = self.horizontal_step_log_proba, self.diagonal_step_log_proba

        def step(prev, i):
            with np.errstate(invalid="ignore"):
                horizontal = expand_many_dims(Hs[:, i], [1, 2, 3]) + prev
                diagonal = expand_many_dims(Ds[:, i], [1, 2]) + prev
            new = logsumexp(horizontal, np.roll(diagonal, 1, axis=3))
            cond = (np.arange(T + 1) <= i).reshape(1, -1, 1, 1)
            return np.where(cond, new, diagonal_gamma)

        fwd = self._unfold(init, step, 1, T)  # [T+1(t2), B, T+1(t1), L, L]
        fwd = np.transpose(fwd, (1, 2, 3, 0, 4))
        mask = np.triu(np.ones((T + 1, T + 1), dtype=bool))
        return apply_logarithmic_mask(fwd, expand_many_dims(mask, [0, 2, 4]))

    def _combine_transition_probabilities(self, a, b):
        """:456-534."""
        B, T, L = self.B, self.T, self.L
        dims_a = a.shape[1:-2]
        dims_b = b.shape[3:]
        a = a.reshape(B, -1, T, L, 1)
        b = b.reshape(B, 1, T, L, -1)
        with np.errstate(invalid="ignore"):
            ab = a + b
            horizontal_blank = expand_many_dims(self._blank_logproba, [1, 3]) + reduce_logsumexp(ab, axis=3)
            act = a + expand_many_dims(self._expected_token_logproba, [1, 4]) + np.roll(b, -1, axis=3)
            diagonal_non_blank = self._select_from_act(act, self._label)
        blank_mask = (np.arange(self.V) == self._blank).reshape(1, 1, 1, -1, 1)
        out = np.where(blank_mask, np.expand_dims(horizontal_blank, 3), diagonal_non_blank)
        return out.reshape((B,) + dims_a + (T, self.V) + dims_b)


LOSS_DATA = {"classic": ClassicCtcLossData, "simplified": SimplifiedCtcLossData}


# ----------------------------------------------------------------------------------------------
# Entry points (base_loss.py:38-99, classic_ctc_loss.py:33-70, simplified_ctc_loss.py:32-67)
# and the chain rule through log-softmax that the reference leaves to TF autodiff.
# ----------------------------------------------------------------------------------------------
def ctc_loss(kind, labels, logits, label_length, logit_length, blank_index=0, dtype=np.float64):
    """Returns the loss-data object built on log_softmax(logits) -- base_loss.py:59-68."""
    logits = np.asarray(logits).astype(dtype)
    return LOSS_DATA[kind](labels, logit_to_logproba(logits, 2), label_length, logit_length, blank_index, dtype)


def classic_ctc_loss(labels, logits, label_length, logit_length, blank_index=0, dtype=np.float64):
    return ctc_loss("classic", labels, logits, label_length, logit_length, blank_index, dtype).loss


def simplified_ctc_loss(labels, logits, label_length, logit_length, blank_index=0, dtype=np.float64):
    return ctc_loss("simplified", labels, logits, label_length, logit_length, blank_index, dtype).loss


def softmax(logits):
    lp = logit_to_logproba(np.asarray(logits), 2)
    return np.exp(lp)


def logits_gradient(data: BaseCtcLossData, logits, d_loss=None):
    """What tape.gradient(loss, logits) returns: TF autodiff of tools.py:37-39 applied to
    d_loss[:,None,None] * gradient (base_loss.py:150-153):
        g_x[t,k] = g[t,k] - softmax(x_t)[k] * sum_k' g[t,k']."""
    g = data.gradient
    if d_loss is not None:
        g = g * np.asarray(d_loss, dtype=g.dtype)[:, None, None]
    s = softmax(np.asarray(logits).astype(g.dtype))
    with np.errstate(invalid="ignore"):
        out = g - s * g.sum(axis=2, keepdims=True)
    # frames whose softmax is ill-defined (all -inf) carry g == 0 there; keep zeros
    return np.where(np.isnan(out), 0.0, out)


def logits_hessian(data: BaseCtcLossData, logits):
    """What tape.batch_jacobian(tape.gradient(sum(loss), logits), logits) returns (README.md:58-71):
        H_x[t1,i,t2,j] = sum_{k1,k2} J[t1,k1,i] H[t1,k1,t2,k2] J[t2,k2,j] - delta_{t1t2} G_t1 (diag(s) - s s^T)[i,j]
    with J[t,k,i] = delta_ki - s_t[i], G_t = sum_k g[t,k]."""
    H = data.hessian  # [B,T,V,T,V]
    g = data.gradient
    s = softmax(np.asarray(logits).astype(H.dtype))
    B, T, V = g.shape
    # right multiply: sum_k2 H[...,t2,k2] (delta_{k2 j} - s[t2,j]) = H[..., t2, j] - (sum_k2 H[...,t2,k2]) s[t2,j]
    Hr = H - H.sum(axis=4, keepdims=True) * s[:, None, None, :, :]
    # left multiply
    Hl = Hr - Hr.sum(axis=2, keepdims=True) * s[:, :, :, None, None]
    G = g.sum(axis=2)  # [B,T]
    for b in range(B):
        for t in range(T):
            Hl[b, t, :, t, :] -= G[b, t] * (np.diag(s[b, t]) - np.outer(s[b, t], s[b, t]))
    return Hl


# ----------------------------------------------------------------------------------------------
# Independent checks (NOT from the reference): brute-force enumeration by the definitions in
# classic_ctc_loss.py:40-52 / simplified_ctc_loss.py:39-49.
# ----------------------------------------------------------------------------------------------
def _collapse(path, blank, kind):
    out = []
    prev = None
    for k in path:
        if kind == "classic":
            if k != blank and k != prev:
                out.append(k)
            prev = k
        else:
            if k != blank:
                out.append(k)
    return tuple(out)


def brute_force(kind, label, logproba, blank=0):
    """Enumerates all V^T paths of ONE sample (fp64).  label: 1-D sequence (already cut to its length),
    logproba: [T, V].  Returns (loss, posterior[T,V], pair[T,V,T,V]) where
    posterior[t,k] = P(path emits k at t | label), pair = joint of two emissions."""
    logproba = np.asarray(logproba, dtype=np.float64)
    T, V = logproba.shape
    target = tuple(int(x) for x in label)
    total = 0.0
    post = np.zeros((T, V))
    pair = np.zeros((T, V, T, V))
    p = np.exp(logproba)
    for path in itertools.product(range(V), repeat=T):
        if _collapse(path, blank, kind) != target:
            continue
        w = 1.0
        for t, k in enumerate(path):
            w *= p[t, k]
        if w == 0.0:
            continue
        total += w
        for t, k in enumerate(path):
            post[t, k] += w
            for t2, k2 in enumerate(path):
                pair[t, k, t2, k2] += w
    if total == 0.0:
        return np.inf, np.zeros((T, V)), np.zeros((T, V, T, V))
    return -np.log(total), post / total, pair / total


# ----------------------------------------------------------------------------------------------
# Deterministic synthetic inputs in the distribution family of tests/common.py:53-104
# (numpy RNG; the reference's tf.random stream is not reproducible without TF).
# ----------------------------------------------------------------------------------------------
def generate_ctc_loss_inputs(batch_size, max_logit_length, random_seed, num_tokens, blank_index=0,
                             max_label_length=None, full_length=False):
    assert blank_index == 0
    rng = np.random.default_rng(random_seed)
    T = max_logit_length
    logits = rng.standard_normal((batch_size, T, num_tokens), dtype=np.float32)
    if max_label_length is None:
        # tests/common.py:77-94 : logit_length ~ U{T//2..T-1}, label_length ~ U{T//4..T//2-1}, labels width T
        logit_length = rng.integers(T // 2, max(T, T // 2 + 1), batch_size, dtype=np.int32)
        label_length = rng.integers(T // 4, max(T // 2, T // 4 + 1), batch_size, dtype=np.int32)
        labels = rng.integers(1, num_tokens, (batch_size, T), dtype=np.int32)
    else:
        U = max_label_length
        if full_length:
            logit_length = np.full(batch_size, T, dtype=np.int32)
            label_length = np.full(batch_size, U, dtype=np.int32)
        else:
            logit_length = rng.integers(T // 2, T, batch_size, dtype=np.int32)
            label_length = rng.integers(U // 2, U + 1, batch_size, dtype=np.int32)
        labels = rng.integers(1, num_tokens, (batch_size, U), dtype=np.int32)
    return dict(labels=labels, logits=logits, label_length=label_length, logit_length=logit_length,
                blank_index=blank_index)
