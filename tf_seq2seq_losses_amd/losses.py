"""Host-side mirror of the reference's Python interface for the CTC hot path.

Same names, argument order, defaults and error behaviour as the reference
(alexeytochin/tf_seq2seq_losses v0.3.0; paths below are relative to it):

    classic_ctc_loss(labels, logits, label_length, logit_length, blank_index=0)      classic_ctc_loss.py:33-70
    simplified_ctc_loss(labels, logits, label_length, logit_length, blank_index=0)   simplified_ctc_loss.py:32-67
    simple_ctc_loss = simplified_ctc_loss       (the name the reference's README/benchmark table uses)
    ctc_loss / ctc_loss_from_logproba                                                 base_loss.py:38-99
    ClassicCtcLossData / SimplifiedCtcLossData  (.loss .gradient .logarithmic_logproba_gradient .hessian
                                                 .alpha .beta)                        base_loss.py:102-298

Tensors are torch tensors on an AMD GPU (TensorFlow is not part of this build); NumPy arrays are accepted
and moved to the current GPU.  Differentiation is wired like the reference's three nested tf.custom_gradient
functions (base_loss.py:140-184): first order = d_loss[:,None,None] * gradient, second order = contraction
with the analytic Hessian, third order raises NotImplementedError.
All arithmetic is done by the HIP kernels behind the C ABI (include/ctc_amd.h); there is no CPU fallback.
"""
from __future__ import annotations

from functools import cached_property
from typing import Optional, Union

import numpy as np
import torch

from . import _lib, ops

TensorLike = Union[torch.Tensor, np.ndarray]


def _as_tensor(x, dtype=None) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x
    if not torch.cuda.is_available():
        raise RuntimeError("tf_seq2seq_losses_amd needs an AMD GPU (no CPU fallback); torch.cuda.is_available() is False")
    t = torch.as_tensor(np.asarray(x))
    if dtype is not None and t.dtype != dtype and not t.dtype.is_floating_point:
        t = t.to(dtype)
    return t.cuda()


def _blank(blank_index) -> int:
    if isinstance(blank_index, torch.Tensor):
        return int(blank_index.item())  # base_loss.py:122-125 accepts a tensor as well
    return int(blank_index)


def _verify_inputs(labels, x, label_length, logit_length):
    """base_loss.py:129-138 : same assertions, same exception type (AssertionError)."""
    assert x.dim() == 3
    # the reference takes float32 only; bfloat16 / float16 activations are an extension
    assert x.dtype in (torch.float32, torch.bfloat16, torch.float16)
    assert labels.dim() == 2
    assert logit_length.dim() == 1
    assert label_length.dim() == 1
    assert x.shape[0] == labels.shape[0]
    assert x.shape[0] == logit_length.shape[0]
    assert x.shape[0] == label_length.shape[0]


# --------------------------------------------------------------------------------------------------
# autograd wiring (base_loss.py:140-184)
# --------------------------------------------------------------------------------------------------
HVP_DENSE = False  # diagnostic: second-order autograd through a materialised [B,T,V,T,V] Hessian, as the reference does


class _HessianContraction(torch.autograd.Function):
    """gradient_fn.backprop (base_loss.py:167-173): out[b,t,k] = sum_{t2,k2} v[b,t2,k2] H[b,t,k,t2,k2].
    Its own backward is the third derivative, which the reference refuses (base_loss.py:179-182)."""

    @staticmethod
    def forward(ctx, x, v, kind, wrt, prep):
        # the reference contracts a materialised [B,T,V,T,V] Hessian here; the tangent-mode kernel (ctc_hvp.hip) gives the
        # same product in O(T*L) memory.  HVP_DENSE keeps the materialised route (parity tests compare both).
        if HVP_DENSE:
            _, _, hess = ops.hessian(kind, wrt, prep, want_grad=False)
            return torch.einsum("btkuj,buj->btk", hess, v.float()).to(x.dtype)
        return ops.hvp(kind, wrt, prep, v)[2].to(x.dtype)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("Third order derivative over the ctc loss function is not implemented.")


class _CtcGradient(torch.autograd.Function):
    """gradient_fn (base_loss.py:157-175) composed with forward_fn.backprop (base_loss.py:150-153):
    returns d_loss[:,None,None] * gradient and differentiates to the Hessian contraction."""

    @staticmethod
    def forward(ctx, x, d_loss, kind, wrt, prep, pending):
        ctx.kind, ctx.wrt, ctx.prep = kind, wrt, prep
        ctx.save_for_backward(x, d_loss)
        if pending is not None:  # the forward pass left its half of the work in a workspace: run the other half (one launch)
            return ops.grad_resume(kind, wrt, prep, pending, d_loss=d_loss)
        return ops.loss_grad(kind, wrt, prep, True, d_loss=d_loss)[1]  # weighting inside the kernel

    @staticmethod
    def backward(ctx, dd):
        x, d_loss = ctx.saved_tensors
        gx = gd = None
        if ctx.needs_input_grad[0]:
            gx = _HessianContraction.apply(x, dd * d_loss.reshape(-1, 1, 1), ctx.kind, ctx.wrt, ctx.prep)
        if ctx.needs_input_grad[1]:
            grad_unit = ops.loss_grad(ctx.kind, ctx.wrt, ctx.prep, True)[1]
            gd = (dd.float() * grad_unit.float()).sum(dim=(1, 2)).to(d_loss.dtype)
        return gx, gd, None, None, None, None


class _CtcLoss(torch.autograd.Function):
    """forward_fn (base_loss.py:140-155).  On the linear-domain fused tier the forward pass runs the first half of the kernel
    (both lattice chains up to their meeting point: the loss) and keeps its checkpoint workspace (64 MB at the north-star
    shape, saved with the graph and released with it); backward runs the second half from there with d_loss applied inside
    the kernel -- together one loss+gradient call's work, no [B,T,V] tensor kept alive in between, no extra pass over the
    gradient for the d_loss weights (ctc_amd_grad_resume).  Other pipelines keep nothing and compute the gradient in one
    call during backward."""

    @staticmethod
    def forward(ctx, x, kind, wrt, prep):
        ctx.kind, ctx.wrt, ctx.prep = kind, wrt, prep
        if x.requires_grad:
            loss, ws = ops.loss_forward(kind, wrt, prep)
        else:
            loss, ws = ops.loss_grad(kind, wrt, prep, want_grad=False)[0], None
        ctx.has_ws = ws is not None
        if ws is not None:
            ctx.save_for_backward(x, ws)  # freed with the graph, like any saved activation
        else:
            ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        x = ctx.saved_tensors[0]
        ws = ctx.saved_tensors[1] if ctx.has_ws else None
        if not torch.is_grad_enabled():
            # first order only (no create_graph): nobody will differentiate the gradient, so skip the nested autograd node
            # (one Function.apply, its context and saved tensors: ~20 us of host time per step)
            if ws is not None:
                return ops.grad_resume(ctx.kind, ctx.wrt, ctx.prep, ws, d_loss=d_loss), None, None, None
            return ops.loss_grad(ctx.kind, ctx.wrt, ctx.prep, True, d_loss=d_loss)[1], None, None, None
        return _CtcGradient.apply(x, d_loss, ctx.kind, ctx.wrt, ctx.prep, ws), None, None, None


def _host_max(label_length):
    """max(label_length) when the caller's copy lives on the host (NumPy array, list, CPU tensor): free, no device sync."""
    if isinstance(label_length, torch.Tensor):
        if label_length.is_cuda or label_length.numel() == 0:
            return None
        return int(label_length.max())
    a = np.asarray(label_length)
    return int(a.max()) if a.size else None


def _ctc(kind_name: str, wrt: int, labels, x, label_length, logit_length, blank_index, max_label_length=None) -> torch.Tensor:
    x = _as_tensor(x)
    labels = _as_tensor(labels, torch.int32)
    if max_label_length is None:
        max_label_length = _host_max(label_length)
    label_length = _as_tensor(label_length, torch.int32)
    logit_length = _as_tensor(logit_length, torch.int32)
    _verify_inputs(labels, x, label_length, logit_length)
    kind = ops.KINDS[kind_name]
    # logits keep the producer's format where the kernels read it directly (bfloat16, time-major views): no copy
    prep = ops.Prepared(labels, x.detach(), label_length, logit_length, _blank(blank_index),
                        keep_format=(wrt == _lib.WRT_LOGITS), host_max_label_length=max_label_length)
    return _CtcLoss.apply(x, kind, wrt, prep)


# --------------------------------------------------------------------------------------------------
# public functions
# --------------------------------------------------------------------------------------------------
def classic_ctc_loss(labels: TensorLike, logits: TensorLike, label_length: TensorLike, logit_length: TensorLike,
                     blank_index: Union[int, torch.Tensor] = 0, *, max_label_length: Optional[int] = None) -> torch.Tensor:
    """Classic CTC loss (repeated tokens without a blank in between are merged, then blanks are dropped;
    reference classic_ctc_loss.py:33-70).  Infeasible samples give loss = +inf with zero gradient.

    Args:
        labels:       [batch, max_label_length] int32
        logits:       [batch, max_length, num_tokens] float32
        label_length: [batch] int32
        logit_length: [batch] int32
        blank_index:  integer >= 0 (or a scalar tensor)
        max_label_length: (extension, keyword only) an upper bound on label_length known on the host.  The reference takes
            the dynamic maximum (base_loss.py:482-486); here any bound >= it gives the same results and a tight one selects
            the fastest kernel tier without a device -> host sync.  Not needed when label_length is passed as a NumPy array /
            CPU tensor (its maximum is then taken on the host), nor for label tensors up to 128 wide.
    Returns: [batch] float32 samplewise loss
    """
    return _ctc("classic", _lib.WRT_LOGITS, labels, logits, label_length, logit_length, blank_index, max_label_length)


def simplified_ctc_loss(labels: TensorLike, logits: TensorLike, label_length: TensorLike, logit_length: TensorLike,
                        blank_index: Union[int, torch.Tensor] = 0, *, max_label_length: Optional[int] = None) -> torch.Tensor:
    """Simplified CTC loss (blanks are dropped, repeated tokens are NOT merged; reference
    simplified_ctc_loss.py:32-67).  Same arguments and return value as classic_ctc_loss."""
    return _ctc("simplified", _lib.WRT_LOGITS, labels, logits, label_length, logit_length, blank_index, max_label_length)


simple_ctc_loss = simplified_ctc_loss  # row label used by the reference's README.md:22 and tests/benchmark.py:72,98


def ctc_loss(labels, logits, label_length, logit_length, blank_index, ctc_loss_data_cls) -> torch.Tensor:
    """base_loss.py:38-68 : generic entry point taking the loss-data class."""
    return _ctc(ctc_loss_data_cls.kind_name, _lib.WRT_LOGITS, labels, logits, label_length, logit_length, blank_index)


def ctc_loss_from_logproba(labels, logprobas, label_length, logit_length, blank_index, ctc_loss_data_cls) -> torch.Tensor:
    """base_loss.py:71-99 : loss as a function of log-probabilities treated as independent variables."""
    return _ctc(ctc_loss_data_cls.kind_name, _lib.WRT_LOGPROBS, labels, logprobas, label_length, logit_length,
                blank_index)


# --------------------------------------------------------------------------------------------------
# loss-data objects (what the reference's unit tests poke at directly)
# --------------------------------------------------------------------------------------------------
class BaseCtcLossData:
    """Mirror of BaseCtcLossData (base_loss.py:102-298) over log-probabilities.  Properties are computed on
    first use by the HIP kernels and memoised, like the reference's cached_property chain."""

    kind_name = ""

    def __init__(self, labels, logprobas, label_length, logit_length, blank_index=0, swap_memory: bool = False):
        logprobas = _as_tensor(logprobas)
        labels = _as_tensor(labels, torch.int32)
        label_length = _as_tensor(label_length, torch.int32)
        logit_length = _as_tensor(logit_length, torch.int32)
        _verify_inputs(labels, logprobas, label_length, logit_length)
        self._kind = ops.KINDS[self.kind_name]
        self._blank_index = _blank(blank_index)
        self._args = (labels, logprobas.detach(), label_length, logit_length)

    @cached_property
    def _max_label_length(self) -> int:
        """base_loss.py:482-486 (dynamic max; costs one device->host sync, only the debug properties use it)."""
        ll = self._args[2]
        return int(ll.max().item()) if ll.numel() else 0

    def _prep(self, exact_u: bool = False) -> ops.Prepared:
        labels, lp, ll, tl = self._args
        return ops.Prepared(labels, lp, ll, tl, self._blank_index, U=self._max_label_length if exact_u else None)

    @cached_property
    def _loss_grad(self):
        return ops.loss_grad(self._kind, _lib.WRT_LOGPROBS, self._prep(), want_grad=True)

    @property
    def loss(self) -> torch.Tensor:
        """[batch]  (classic_ctc_loss.py:152-165, simplified_ctc_loss.py:73-83)"""
        return self._loss_grad[0]

    @property
    def gradient(self) -> torch.Tensor:
        """[batch, max_logit_length, num_tokens]: d loss / d logproba = -posterior (base_loss.py:262-268)"""
        return self._loss_grad[1]

    @cached_property
    def logarithmic_logproba_gradient(self) -> torch.Tensor:
        """log of the posterior = log(-gradient), computed in log space like the reference does (base_loss.py:270-298): finite
        where the float32 gradient underflows (e^-150 comes out as -150); -inf on padded frames, for infeasible samples and
        for tokens no lattice state emits."""
        return ops.log_posterior(self._kind, _lib.WRT_LOGPROBS, self._prep())[1]

    @cached_property
    def _alpha_beta(self):
        return ops.alpha_beta(self._kind, _lib.WRT_LOGPROBS, self._prep(exact_u=True))

    @property
    def alpha(self) -> torch.Tensor:
        """classic [batch, T+1, U+1, 2], simplified [batch, T+1, U+1]; natural log, -inf = impossible"""
        return self._alpha_beta[1]

    @property
    def beta(self) -> torch.Tensor:
        return self._alpha_beta[2]

    @cached_property
    def hessian(self) -> torch.Tensor:
        """[batch, T, V, T, V]: second derivative w.r.t. log-probabilities (base_loss.py:186-260)"""
        return ops.hessian(self._kind, _lib.WRT_LOGPROBS, self._prep(), want_grad=False)[2]

    def hessian_vector_product(self, vec: TensorLike) -> torch.Tensor:
        """sum_{t2,k2} hessian[b,t,k,t2,k2] vec[b,t2,k2] without building `hessian` (extension: tangent-mode
        alpha/beta, ctc_amd_hvp); equals torch.einsum("btkuj,buj->btk", self.hessian, vec)."""
        return ops.hvp(self._kind, _lib.WRT_LOGPROBS, self._prep(), _as_tensor(vec))[2]

    @property
    def gamma(self):
        raise NotImplementedError(
            "gamma (classic_ctc_loss.py:167-308, simplified_ctc_loss.py:85-191) is an O(T^2 L^2) intermediate "
            "of the reference's Hessian; this implementation never materialises it (see DESIGN.md).")


class ClassicCtcLossData(BaseCtcLossData):
    kind_name = "classic"


class SimplifiedCtcLossData(BaseCtcLossData):
    kind_name = "simplified"
