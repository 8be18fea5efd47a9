"""Diagnostic: where the time of the autograd wrapper goes at the reference's benchmark shape (B=256, T=255, V=32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tf_seq2seq_losses_amd as ctc
from benchmarks.reference_table import make_inputs
dev = torch.device("cuda:0")
labels, logits, ll, tl = make_inputs(256, 255, 32, 0, dev)


def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


x = logits.detach().requires_grad_(True)
print("forward (requires_grad)      %.3f ms" % t(lambda: ctc.classic_ctc_loss(labels, x, ll, tl, 0)))
print("forward (no grad)            %.3f ms" % t(lambda: ctc.classic_ctc_loss(labels, logits, ll, tl, 0)))
def fb():
    loss = ctc.classic_ctc_loss(labels, x, ll, tl, 0)
    return torch.autograd.grad(loss.sum(), x)[0]
print("forward + grad(loss.sum())   %.3f ms" % t(fb))
def fb2():
    loss = ctc.classic_ctc_loss(labels, x, ll, tl, 0)
    return torch.autograd.grad(loss[torch.isfinite(loss)].sum(), x)[0]
print("forward + grad(masked sum)   %.3f ms" % t(fb2))
from tf_seq2seq_losses_amd import ops, _lib
p = ops.Prepared(labels, logits, ll, tl, 0)
print("ops.Prepared                 %.3f ms" % t(lambda: ops.Prepared(labels, logits, ll, tl, 0)))
print("ops.loss_grad (C ABI + alloc) %.3f ms" % t(lambda: ops.loss_grad(0, 0, p, True)))
print("pipeline", _lib.pipeline_name(0, 0, p.B, p.T, p.V, p.U, True), "U", p.U)
