"""Batch-sharded multi-GPU use of the CTC loss: one process per GPU (torch.distributed, backend "nccl" = RCCL
on ROCm, "gloo" in CPU tests).

The reference has no distributed code at all.  The path shards trivially: every tensor has a
leading batch axis and no operation mixes batch entries (the only cross-batch construct in the reference is the
segment-id offset b*V that keeps segments disjoint, base_loss.py:450-455).  So each rank runs the kernels on its
own contiguous block of utterances with NO data-path collective; gradients are w.r.t. activations (logits) and stay
sharded.  The single collective is the all-reduce of the scalar the training loop takes from the loss
(tf.reduce_sum / reduce_mean in README.md:62, tests/benchmark.py:199): 8 bytes, latency-bound on xGMI.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of a global batch owned by `rank`; the first batch % world ranks get one extra."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_loss_sum(local_loss: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-reduces (sum of finite losses, number of finite losses) over the ranks: ONE kernel launch (ctc_amd_reduce_loss)
    and ONE collective of two floats.  Returns (global_sum, global_count).  `async_op=True` in all_reduce_pair below gives
    the handle instead, for loops that read the scalar a step later (bench.py)."""
    buf, work = all_reduce_pair(local_loss, group=group, async_op=False)
    return buf[0], buf[1]


def local_pair(local_loss: torch.Tensor) -> torch.Tensor:
    """[sum of finite losses, number of finite losses] of this rank, float32[2] on the loss's device (float32 whatever the
    dtype of the losses: the pair is what gets all-reduced)."""
    if local_loss.is_cuda:
        from . import _lib, ops
        buf = torch.empty(2, dtype=torch.float32, device=local_loss.device)
        x = local_loss if (local_loss.dtype == torch.float32 and local_loss.is_contiguous()) else local_loss.float().contiguous()
        with ops._on_device(x.device):  # (the loss may live on a device that is not the current one)
            _lib.check(_lib.load().ctc_amd_reduce_loss(x.data_ptr() if x.numel() else None, x.numel(), buf.data_ptr(),
                                                       ops._stream(x.device)), "ctc_amd_reduce_loss")
        return buf
    finite = torch.isfinite(local_loss)  # CPU tensors (gloo tests): plain torch
    return torch.stack([torch.where(finite, local_loss, torch.zeros_like(local_loss)).sum(), finite.sum().to(local_loss.dtype)]).float()


def all_reduce_pair(local_loss: torch.Tensor, group=None, async_op: bool = False):
    """(buffer, work): the [sum, count] pair of this rank, all-reduced in place over the group (work is None when there is
    nothing to wait for)."""
    buf = local_pair(local_loss)
    work = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return buf, (work if async_op else None)


def pipelined_steps(step: Callable[[], torch.Tensor], steps: int, group=None, reduced: bool = False,
                    consume: Callable[[int, torch.Tensor], None] = None, depth: int = 1,
                    all_reduce: Callable = None, every: int = 1, group_view: Callable[[int, int], torch.Tensor] = None):
    """The data-parallel loop of bench.py: every step computes this rank's losses and issues the all-reduce of its
    [sum, count] pair asynchronously; the pair of step i-depth is waited for after step i has been launched, so the
    collective (two numbers, latency-bound) runs beside the next `depth` kernels.  Returns the list of reduced pairs, all
    complete on return.

    reduced=True: `step()` returns this rank's pair itself -- the int64[2] buffer ctc_amd_loss_grad_sum accumulated inside
    the loss kernel (fixed point: the all-reduced total has the same bits whatever the order) -- and no reduction launch is
    left on the stream.  The caller cycles depth + 2 such buffers: step i fills buffer i mod (depth + 2) and clears the next
    one, last used by step i - depth - 1, whose collective has been waited for by then.  `consume(i, pair)` is called once the
    pair of step i is complete and before its buffer is recycled; in this mode the returned list is empty.

    all_reduce: stand-in for dist.all_reduce(buf, async_op=True) returning an object with .wait() (bench.py
    --emulate-collective measures the loop on one GPU with a kernel of RCCL's footprint); default: the real collective when a
    process group of more than one rank exists.

    every > 1 (reduced=True only): fewer, larger collectives -- the pairs of `every` consecutive steps go out in ONE all-reduce
    issued after the last of them (the last group may be shorter).  The caller keeps the pairs in one int64[(depth + 2) * every, 2]
    tensor, step i filling row i mod ((depth + 2) * every) and clearing the next one, and `group_view(first_step, n)` returns
    the contiguous rows of steps first_step .. first_step + n - 1; `depth` then counts groups in flight and `consume(i, rows)`
    is called once per group with the index of its last step.  What each collective costs the compute stream is the event
    record / stream wait pair around it (ProcessGroupNCCL's ordering), not its 16 bytes: bench.py's emulated-collective table."""
    assert depth >= 1 and every >= 1
    assert every == 1 or (reduced and group_view is not None)
    from collections import deque
    out, pending = [], deque()

    def issue(buf):
        if all_reduce is not None:
            return all_reduce(buf)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
        return None

    def retire():
        buf, work, i = pending.popleft()
        if work is not None:
            work.wait()
        if consume is not None:
            consume(i, buf)

    for i in range(steps):
        if reduced:
            buf = step()
        else:
            buf = local_pair(step())
        if every > 1:
            if i % every != every - 1 and i != steps - 1:
                continue
            first = i - i % every
            buf = group_view(first, i - first + 1)
        pending.append((buf, issue(buf), i))
        if len(pending) > depth:
            retire()
        if not reduced:
            out.append(buf)
    while pending:
        retire()
    return out


def sharded_loss(loss_fn: Callable[..., torch.Tensor], labels, logits, label_length, logit_length, blank_index=0,
                 group=None):
    """Runs `loss_fn` (classic_ctc_loss / simplified_ctc_loss) on this rank's shard of a replicated global batch
    and returns (local_loss[shard], global_sum, global_count).  Mostly a convenience for tests and examples: in
    training each rank normally already holds only its own shard."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(int(logits.shape[0]), rank, world)
    local = loss_fn(labels[lo:hi], logits[lo:hi], label_length[lo:hi], logit_length[lo:hi], blank_index)
    s, n = reduce_loss_sum(local, group)
    return local, s, n
