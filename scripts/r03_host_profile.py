"""Diagnostic: host time of the drop-in Python path at the reference's benchmark shape (B=256, T=255, V=32, label tensor 255
wide): wall clock per call of forward and forward+gradient, and a cProfile of 2000 forward+gradient calls (top functions by
own time).  usage: python scripts/r03_host_profile.py > gpurun_out/r03_host_profile.txt"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tf_seq2seq_losses_amd as ctc
from benchmarks.reference_table import make_inputs

dev = torch.device("cuda:0")
labels, logits, ll, tl = make_inputs(256, 255, 32, 0, dev)


def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def submit(fn, n=300):
    """host time to SUBMIT a call (no sync inside the loop; the queue is drained first)"""
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize(); return dt


x = logits.detach().requires_grad_(True)
def fwd():
    with torch.no_grad():
        return ctc.classic_ctc_loss(labels, logits, ll, tl, 0)
def fb():
    loss = ctc.classic_ctc_loss(labels, x, ll, tl, 0)
    return torch.autograd.grad(loss.sum(), x)[0]
def fb_ref():  # the exact recipe of benchmarks/reference_table.py
    xx = logits.detach().requires_grad_(True)
    loss = ctc.classic_ctc_loss(labels, xx, ll, tl, 0)
    return torch.autograd.grad(torch.where(torch.isfinite(loss), loss, 0.0).sum(), xx)[0]
ll_h = ll.cpu()
def fb_hint():
    loss = ctc.classic_ctc_loss(labels, x, ll, tl, 0, max_label_length=126)
    return torch.autograd.grad(loss.sum(), x)[0]

from tf_seq2seq_losses_amd import ops, _lib
for name, kw in (("default", {}), ("hinted", dict(host_max_label_length=126))):
    p = ops.Prepared(labels, logits, ll, tl, 0, **kw)
    print(name, "U", p.U, "pipeline", _lib.pipeline_name(0, 0, p.B, p.T, p.V, p.U, True), "ws", _lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, p.B, p.T, p.V, p.U))
assert torch.equal(fb(), fb_hint())


def dev_us(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for rep in range(2):
    for name, fn in (("fb", fb), ("fb_hint", fb_hint), ("fb_ref", fb_ref), ("fwd", fwd)):
        print("rep %d %-8s wall %.4f ms  submit %.4f ms  device %.1f us per call" % (rep, name, t(fn), submit(fn), dev_us(fn)))
print("forward (no grad)               %.4f ms   submit %.4f" % (t(fwd), submit(fwd)))
print("forward + grad(loss.sum())      %.4f ms   submit %.4f" % (t(fb), submit(fb)))
print("forward + grad, table recipe    %.4f ms   submit %.4f" % (t(fb_ref), submit(fb_ref)))
print("forward + grad, hinted          %.4f ms   submit %.4f" % (t(fb_hint), submit(fb_hint)))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    fb()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
