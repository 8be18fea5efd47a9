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
    """All-reduces (sum of finite losses, number of finite losses) over the ranks: ONE collective of two floats.
    Returns (global_sum, global_count)."""
    finite = torch.isfinite(local_loss)
    buf = torch.stack([torch.where(finite, local_loss, torch.zeros_like(local_loss)).sum(),
                       finite.sum().to(local_loss.dtype)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[0], buf[1]


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
