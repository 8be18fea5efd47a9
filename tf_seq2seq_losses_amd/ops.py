"""Thin torch<->C-ABI plumbing: device pointers, the current HIP stream and workspace allocation.
torch is used for device memory and streams only; all arithmetic happens in libctc_amd.so."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

KINDS = {"classic": _lib.CLASSIC, "simplified": _lib.SIMPLIFIED}
# Label tensors wider than this are worth one look at max(label_length): up to 128 positions run the fastest instantiation
# (two positions per lane) whatever the width says; beyond, every doubling selects a slower tier (four / eight positions per
# lane, then the three-kernel pipeline) although the labels inside may be short (tests/common.py:89-94 pads labels to T).
WIDTH_WORTH_A_LOOK = 128
_MAXLEN_CACHE = {}  # id(label_length tensor) -> (weak reference to it, its version, its maximum): one sync per distinct tensor


def _device_max_label_length(t: torch.Tensor) -> int:
    """max(label_length) of a device tensor, fetched once per tensor OBJECT and version (a data pointer is no identity: the
    caching allocator hands the same address to the next batch's tensor)."""
    import weakref
    hit = _MAXLEN_CACHE.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    if len(_MAXLEN_CACHE) > 64:
        _MAXLEN_CACHE.clear()
    m = int(t.max().item())  # device -> host sync (pass max_label_length= to avoid it)
    _MAXLEN_CACHE[id(t)] = (weakref.ref(t), t._version, m)
    return m


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            "tf_seq2seq_losses_amd runs on an AMD GPU only (HIP kernels for gfx950); got a CPU tensor and "
            "there is no CPU fallback. Move the inputs to the GPU (tensor.cuda()).")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


class Prepared:
    """Validated device inputs of one call.  By default the logits are made contiguous float32 [B,T,V]; with
    keep_format=True a float32/bfloat16 tensor whose token axis is contiguous is passed as it is (time-major views,
    bfloat16 activations: ctc_amd_loss_grad_ex) -- only loss_grad takes such inputs, the other entry points use plain()."""

    def __init__(self, labels, x, label_length, logit_length, blank_index, U=None, keep_format=False, host_max_label_length=None):
        _require_gpu(x)
        dev = x.device
        V = int(x.shape[2]) if x.dim() == 3 else 0
        native = (keep_format and x.dim() == 3 and x.dtype in _DTYPES and x.numel() > 0 and x.stride(2) == 1
                  and x.stride(0) >= V and x.stride(1) >= V)
        self.native = bool(native) and not (x.dtype == torch.float32 and x.is_contiguous())
        self.x = x if (native or (x.dtype == torch.float32 and x.is_contiguous())) else x.to(torch.float32).contiguous()
        def i32(t):
            if t.dtype == torch.int32 and t.device == dev and t.is_contiguous():
                return t
            return t.to(device=dev, dtype=torch.int32).contiguous()
        src_label_length = label_length  # (the caller's own tensor: what the cache of its maximum is keyed on)
        self.labels, self.label_length, self.logit_length = i32(labels), i32(label_length), i32(logit_length)
        self.blank = int(blank_index)
        self.B, self.T, self.V = (int(s) for s in x.shape)
        self.stride = int(self.labels.shape[1])
        # static bound on the label length: any U >= max(label_length) gives the reference's results (it uses the dynamic
        # maximum, base_loss.py:482-486).  The width of the label tensor always works; a tighter bound selects a faster kernel
        # tier.  Sources, in order: the caller's hint (`max_label_length=` of the public functions), the host copy of
        # label_length when the caller passed one (free), and -- only for tensors wide enough that it matters -- the
        # device-side maximum, fetched once per distinct label_length tensor (one sync, not capturable into a graph).
        if U is None:
            U = self.stride
            if host_max_label_length is not None:
                U = max(0, min(U, int(host_max_label_length)))
            elif U > WIDTH_WORTH_A_LOOK and self.label_length.numel() > 0:
                # the maximum of the CALLER's tensor: a host tensor costs nothing; a device tensor one sync per tensor object and
                # version (keyed on the caller's object -- the converted copy made above is a new object every call and never hit
                # the cache); under stream capture no sync is possible and the width stands
                if not src_label_length.is_cuda:
                    U = max(0, min(U, int(src_label_length.max())))
                elif not torch.cuda.is_current_stream_capturing():
                    U = max(0, min(U, _device_max_label_length(src_label_length)))
        self.U = int(U)
        self.device = dev

    def plain(self) -> "Prepared":
        """The same inputs with contiguous float32 logits (Hessian, HVP and alpha/beta read that format only), 16-byte aligned
        (ctc_amd_hvp / ctc_amd_hessian refuse other base pointers: a batch-sliced view whose offset is not a multiple of 16 bytes is
        copied once here)."""
        if not self.native:
            if self.x.numel() > 0 and (self.x.data_ptr() & 15) != 0:
                q = Prepared.__new__(Prepared)
                q.__dict__.update(self.__dict__)
                q.x = self.x.clone()
                return q
            return self
        q = Prepared.__new__(Prepared)
        q.__dict__.update(self.__dict__)
        q.x = self.x.to(torch.float32).contiguous()
        q.native = False
        return q

    def common(self, kind: int, wrt: int):
        assert not self.native, "this entry point takes contiguous float32 logits: use Prepared.plain()"
        return (kind, wrt, _ptr(self.x), _ptr(self.labels), self.stride, _ptr(self.label_length),
                _ptr(self.logit_length), self.blank, self.B, self.T, self.V, self.U)


_WS_BYTES = {}   # (what, kind, B, T, V, U) -> bytes: no ctypes round trip per call
_WS_CACHE = {}   # (device, stream) -> the loss+gradient workspace last used there (reused while it is large enough)
_PIPELINE = {}   # (kind, wrt, B, T, V, U) -> pipeline name of a contiguous float32 call


def _ws_bytes(what: int, kind: int, p: "Prepared") -> int:
    key = (what, kind, p.B, p.T, p.V, p.U, _lib.override_generation)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = _lib.workspace_bytes(what, kind, p.B, p.T, p.V, p.U)
    return n


def _loss_grad_selector(wrt: int, p: "Prepared") -> int:
    """Workspace selector of a loss+gradient call: the pipeline's own (small) layout for logits input in a format the
    fused tiers read (float32; bfloat16 with 8-byte aligned rows), the conservative one otherwise (include/ctc_amd.h)."""
    if wrt != _lib.WRT_LOGITS:
        return _lib.WS_LOSS_GRAD
    x = p.x
    if x.dtype == torch.bfloat16 and ((p.V | x.stride(0) | x.stride(1)) & 3 or x.data_ptr() & 7):
        return _lib.WS_LOSS_GRAD
    if x.dtype == torch.float16:  # read by the three-kernel pipeline only
        return _lib.WS_LOSS_GRAD
    return _lib.WS_LOSS_GRAD_LOGITS


def pipeline_of(kind: int, wrt: int, p: "Prepared") -> str:
    key = (kind, wrt, p.B, p.T, p.V, p.U, _lib.override_generation)
    name = _PIPELINE.get(key)
    if name is None:
        name = _PIPELINE[key] = _lib.pipeline_name(kind, wrt, p.B, p.T, p.V, p.U, True)
    return name


def _workspace(what: int, kind: int, p: Prepared) -> torch.Tensor:
    n = _ws_bytes(what, kind, p)
    if what not in (_lib.WS_LOSS_GRAD, _lib.WS_LOSS_GRAD_LOGITS):
        return torch.empty(max(n, 1), dtype=torch.uint8, device=p.device)
    # The loss+gradient workspace is scratch that lives only for the duration of one call.  Calls on one stream are
    # ordered, so they can share one buffer; another stream gets its own.
    ck = (p.device, _stream(p.device))
    ws = _WS_CACHE.get(ck)
    if ws is None or ws.numel() < n:
        ws = _WS_CACHE[ck] = torch.empty(max(n, 1), dtype=torch.uint8, device=p.device)
    return ws


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev) -> int:
    """hipStream_t of torch's current stream on `dev` (torch.cuda.current_stream builds a Stream object: ~15 us per call)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` costs ~10 us of host time per call; most calls already run on the right device."""

    __slots__ = ("dev", "ctx")

    def __init__(self, dev):
        self.dev, self.ctx = dev, None

    def __enter__(self):
        if self.dev.index is not None and torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def loss_grad(kind: int, wrt: int, p: Prepared, want_grad: bool, d_loss: Optional[torch.Tensor] = None,
              workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    lib = _lib.load()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    if p.native:  # the gradient goes back in the producer's format: same element type, same strides
        grad = torch.empty_strided(p.x.shape, p.x.stride(), dtype=p.x.dtype, device=p.device) if want_grad else None
    else:
        grad = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device) if want_grad else None
    if p.B == 0:
        return loss, grad
    ws = workspace if workspace is not None else _workspace(_loss_grad_selector(wrt, p), kind, p)
    if d_loss is not None:
        d_loss = d_loss.to(device=p.device, dtype=torch.float32).contiguous()
    with _on_device(p.device):
        if p.native:
            dt = _DTYPES[p.x.dtype]
            rc = lib.ctc_amd_loss_grad_ex(kind, wrt, _ptr(p.x), dt, p.x.stride(0), p.x.stride(1), _ptr(p.labels), p.stride,
                                          _ptr(p.label_length), _ptr(p.logit_length), p.blank, p.B, p.T, p.V, p.U,
                                          _ptr(loss), _ptr(grad), dt, p.x.stride(0), p.x.stride(1), _ptr(d_loss),
                                          ws.data_ptr(), ws.numel(), _stream(p.device))
        else:
            rc = lib.ctc_amd_loss_grad(*p.common(kind, wrt), _ptr(loss), _ptr(grad), _ptr(d_loss),
                                       ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_loss_grad")
    return loss, grad


def loss_grad_packed(kind: int, wrt: int, labels: torch.Tensor, x: torch.Tensor, row_offsets: torch.Tensor,
                     label_length: torch.Tensor, logit_length: torch.Tensor, blank_index: int, max_length: int,
                     U: Optional[int] = None, want_grad: bool = True, d_loss: Optional[torch.Tensor] = None):
    """Packed (ragged) batch, ctc_amd_loss_grad_packed: `x` is [total_rows, V] (float32 / bfloat16 / float16, rows contiguous
    in the token axis), utterance b owns the rows row_offsets[b] .. row_offsets[b] + logit_length[b] - 1; the gradient comes
    back with the same packing and element type.  max_length >= max(logit_length) sizes the workspace."""
    lib = _lib.load()
    _require_gpu(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in _DTYPES
    dev = x.device
    B, V, T = int(labels.shape[0]), int(x.shape[1]), int(max_length)
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
    labels, label_length, logit_length = i32(labels), i32(label_length), i32(logit_length)
    row_offsets = row_offsets.to(device=dev, dtype=torch.int64).contiguous()
    U = int(labels.shape[1]) if U is None else int(U)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    grad = torch.zeros_like(x) if want_grad else None  # (rows no utterance owns stay zero)
    if B == 0:
        return loss, grad
    ws = torch.empty(max(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, kind, B, T, V, U), 1), dtype=torch.uint8, device=dev)
    if d_loss is not None:
        d_loss = d_loss.to(device=dev, dtype=torch.float32).contiguous()
    dt = _DTYPES[x.dtype]
    with _on_device(dev):
        rc = lib.ctc_amd_loss_grad_packed(kind, wrt, _ptr(x), dt, _ptr(row_offsets), x.stride(0), _ptr(labels), int(labels.shape[1]),
                                          _ptr(label_length), _ptr(logit_length), int(blank_index), B, T, V, U, _ptr(loss),
                                          _ptr(grad), dt, grad.stride(0) if grad is not None else V, _ptr(d_loss),
                                          ws.data_ptr(), ws.numel(), _stream(dev))
    _lib.check(rc, "ctc_amd_loss_grad_packed")
    return loss, grad


LOSS_SUM_SCALE = 1.0 / 1048576.0  # unit of the fixed-point loss sum (ctc_amd_loss_grad_sum)


def loss_grad_sum(kind: int, wrt: int, p: Prepared, sum2: torch.Tensor, zero_next: Optional[torch.Tensor] = None,
                  want_grad: bool = True, d_loss: Optional[torch.Tensor] = None):
    """loss_grad that also adds [sum of the finite losses in units of 2^-20, their number] to the int64[2] tensor `sum2`
    inside the same launch (exact integer adds: deterministic) and clears `zero_next` for the following step."""
    lib = _lib.load()
    assert sum2.dtype == torch.int64 and sum2.numel() == 2 and sum2.device == p.device
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    if p.native:
        grad = torch.empty_strided(p.x.shape, p.x.stride(), dtype=p.x.dtype, device=p.device) if want_grad else None
    else:
        grad = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device) if want_grad else None
    ws = _workspace(_loss_grad_selector(wrt, p), kind, p)
    if d_loss is not None:
        d_loss = d_loss.to(device=p.device, dtype=torch.float32).contiguous()
    dt = _DTYPES[p.x.dtype]
    gs = (grad.stride(0), grad.stride(1)) if grad is not None else (p.T * p.V, p.V)
    with _on_device(p.device):
        rc = lib.ctc_amd_loss_grad_sum(kind, wrt, _ptr(p.x), dt, p.x.stride(0), p.x.stride(1), _ptr(p.labels), p.stride,
                                       _ptr(p.label_length), _ptr(p.logit_length), p.blank, p.B, p.T, p.V, p.U,
                                       _ptr(loss), _ptr(grad), dt, gs[0], gs[1], _ptr(d_loss), _ptr(sum2), _ptr(zero_next),
                                       ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_loss_grad_sum")
    return loss, grad


def loss_forward(kind: int, wrt: int, p: Prepared, keep_always: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Loss only.  Where the two-call form exists (the linear-domain fused tier) the call gets a workspace of its own that
    grad_resume continues from (returned; keep it alive until then: checkpoint rows only, ~64 MB at the north-star shape);
    every other pipeline would recompute everything in the second call anyway, so nothing is kept (None) unless
    keep_always (the parity tests call ctc_amd_grad_resume behind every pipeline)."""
    sel = _loss_grad_selector(wrt, p)
    if p.B == 0 or (not keep_always and (sel != _lib.WS_LOSS_GRAD_LOGITS or pipeline_of(kind, wrt, p) != "fused6")):
        return loss_grad(kind, wrt, p, False)[0], None
    ws = torch.empty(max(_ws_bytes(sel, kind, p), 1), dtype=torch.uint8, device=p.device)
    if keep_always and pipeline_of(kind, wrt, p) != "fused6":
        loss, _ = loss_grad(kind, wrt, p, False, workspace=ws)
        return loss, ws
    # first half of a forward / backward pair (ctc_amd_loss_forward, ABI v5): the resume call verifies every utterance's posterior
    # mass, so the linear-domain kernel keeps its conservative loss-only signs for binding alignments only
    lib = _lib.load()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    with _on_device(p.device):
        rc = lib.ctc_amd_loss_forward(kind, wrt, _ptr(p.x), _DTYPES[p.x.dtype], p.x.stride(0), p.x.stride(1), _ptr(p.labels), p.stride,
                                      _ptr(p.label_length), _ptr(p.logit_length), p.blank, p.B, p.T, p.V, p.U,
                                      _ptr(loss), ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_loss_forward")
    return loss, ws


def grad_resume(kind: int, wrt: int, p: Prepared, ws: torch.Tensor, d_loss: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient for the loss that loss_forward computed into `ws` (ctc_amd_grad_resume), weighted by d_loss.  The losses the
    call rewrites (utterances redone in the log domain) go to a scratch buffer: the tensor the forward pass returned to the
    user is never written again."""
    if ws is None:  # the forward call kept nothing (its pipeline has no two-call form): one loss+gradient call
        return loss_grad(kind, wrt, p, True, d_loss=d_loss)[1]
    lib = _lib.load()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    if p.native:
        grad = torch.empty_strided(p.x.shape, p.x.stride(), dtype=p.x.dtype, device=p.device)
    else:
        grad = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device)
    if p.B == 0 or p.T == 0:
        return grad
    if d_loss is not None:
        d_loss = d_loss.to(device=p.device, dtype=torch.float32).contiguous()
    dt = _DTYPES[p.x.dtype]
    with _on_device(p.device):
        rc = lib.ctc_amd_grad_resume(kind, wrt, _ptr(p.x), dt, p.x.stride(0), p.x.stride(1), _ptr(p.labels), p.stride,
                                     _ptr(p.label_length), _ptr(p.logit_length), p.blank, p.B, p.T, p.V, p.U,
                                     _ptr(loss), _ptr(grad), dt, grad.stride(0), grad.stride(1), _ptr(d_loss),
                                     ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_grad_resume")
    return grad


def fused_flags(ws: torch.Tensor, kind: int, p: Prepared) -> torch.Tensor:
    """Diagnostic: int32[B] flag words the linear-domain kernel left in `ws` (0 = linear domain, else redone in the log domain)."""
    off = _lib.flags_offset(kind, p.B, p.T, p.V, p.U)
    return ws[off:off + 4 * p.B].view(torch.int32)


def alpha_beta(kind: int, wrt: int, p: Prepared):
    lib = _lib.load()
    p = p.plain()
    L = p.U + 1
    shape = (p.B, p.T + 1, L, 2) if kind == _lib.CLASSIC else (p.B, p.T + 1, L)
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    alpha = torch.empty(shape, dtype=torch.float32, device=p.device)
    beta = torch.empty(shape, dtype=torch.float32, device=p.device)
    if p.B == 0:
        return loss, alpha, beta
    ws = _workspace(_lib.WS_ALPHA_BETA, kind, p)
    with _on_device(p.device):
        rc = lib.ctc_amd_alpha_beta(*p.common(kind, wrt), _ptr(loss), _ptr(alpha), _ptr(beta),
                                    ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_alpha_beta")
    return loss, alpha, beta


def log_posterior(kind: int, wrt: int, p: Prepared):
    """(loss[B], lg[B,T,V]): natural log of the posterior of "frame t emits token k", in log space (ctc_amd_log_posterior)."""
    lib = _lib.load()
    p = p.plain()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    lg = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device)
    if p.B == 0:
        return loss, lg
    ws = _workspace(_lib.WS_ALPHA_BETA, kind, p)
    with _on_device(p.device):
        rc = lib.ctc_amd_log_posterior(*p.common(kind, wrt), _ptr(loss), _ptr(lg), ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_log_posterior")
    return loss, lg


def hessian(kind: int, wrt: int, p: Prepared, want_grad: bool = True):
    lib = _lib.load()
    p = p.plain()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    grad = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device) if want_grad else None
    hess = torch.empty((p.B, p.T, p.V, p.T, p.V), dtype=torch.float32, device=p.device)
    if p.B == 0 or p.T == 0:
        if p.B and not p.T:
            loss, _ = loss_grad(kind, wrt, p, False)
        return loss, grad, hess
    ws = _workspace(_lib.WS_HESSIAN, kind, p)
    with _on_device(p.device):
        rc = lib.ctc_amd_hessian(*p.common(kind, wrt), _ptr(loss), _ptr(grad), _ptr(hess),
                                 ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_hessian")
    return loss, grad, hess


def hvp(kind: int, wrt: int, p: Prepared, vec: torch.Tensor, want_grad: bool = False, return_workspace: bool = False):
    """out[b,t,k] = sum_{t2,k2} H[b,t,k,t2,k2] vec[b,t2,k2] through ctc_amd_hvp (no [B,T,V,T,V] tensor).
    return_workspace (diagnostics): also the workspace, whose flag words say which utterances the fused kernel redid in the log
    domain (_lib.hvp_flags_offset)."""
    lib = _lib.load()
    p = p.plain()
    assert tuple(vec.shape) == (p.B, p.T, p.V), f"vec must be [B,T,V] = {(p.B, p.T, p.V)}, got {tuple(vec.shape)}"
    vec = vec.to(device=p.device, dtype=torch.float32).contiguous()
    if vec.numel() > 0 and (vec.data_ptr() & 15) != 0:  # (an offset view: ctc_amd_hvp wants 16-byte aligned rows)
        vec = vec.clone()
    loss = torch.empty(p.B, dtype=torch.float32, device=p.device)
    grad = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device) if want_grad else None
    out = torch.empty((p.B, p.T, p.V), dtype=torch.float32, device=p.device)
    if p.B == 0 or p.T == 0:
        if p.B and not p.T:
            loss, _ = loss_grad(kind, wrt, p, False)
        return loss, grad, out
    ws = _workspace(_lib.WS_HVP, kind, p)
    with _on_device(p.device):
        rc = lib.ctc_amd_hvp(*p.common(kind, wrt), _ptr(vec), _ptr(loss), _ptr(grad), _ptr(out),
                             ws.data_ptr(), ws.numel(), _stream(p.device))
    _lib.check(rc, "ctc_amd_hvp")
    if return_workspace:
        return loss, grad, out, ws
    return loss, grad, out


def check_labels(labels, label_length, num_tokens: int, blank_index: int = 0) -> None:
    """Opt-in validation (off the hot path: synchronises): raises ValueError if a label inside its `label_length` lies
    outside [0, num_tokens) or equals `blank_index` -- what TF-CPU's gather reports as InvalidArgumentError for
    out-of-range labels (base_loss.py:328-344).  The loss functions themselves treat such a label as an impossible
    emission (loss +inf, zero gradient)."""
    labels = torch.as_tensor(labels)
    label_length = torch.as_tensor(label_length)
    if not labels.is_cuda:
        labels = labels.cuda()
    label_length = label_length.to(labels.device)
    labels = labels.to(torch.int32).contiguous()
    label_length = label_length.to(torch.int32).contiguous()
    B, U = int(labels.shape[0]), int(labels.shape[1])
    if B == 0 or U == 0:
        return
    with _on_device(labels.device):
        _lib.check(_lib.load().ctc_amd_check_labels(_ptr(labels), U, _ptr(label_length), int(blank_index), B, int(num_tokens), U,
                                                    _stream(labels.device)), "ctc_amd_check_labels")
