"""Diagnostic: step = loss+grad call + ctc_amd_reduce_loss launch  vs  step = ctc_amd_loss_grad_sum (sum inside the launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops, dist as cdist
B, T, U, V = 256, 1000, 128, 256
dev = torch.device("cuda:0")
host, d = bench.make_inputs(B, T, U, V, 0, False, dev)
p = ops.Prepared(d["labels"], d["logits"], d["label_length"], d["logit_length"], 0, U=U)
lib = _lib.load()
ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, B, T, V, U), dtype=torch.uint8, device=dev)
loss = torch.empty(B, device=dev); grad = torch.empty(B, T, V, device=dev)
sums = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(3)]
out2 = torch.empty(2, device=dev)
st = torch.cuda.current_stream().cuda_stream
a1 = p.common(0, 0) + (loss.data_ptr(), grad.data_ptr(), None, ws.data_ptr(), ws.numel(), st)
x = d["logits"]
a2 = (0, 0, x.data_ptr(), 0, x.stride(0), x.stride(1), p.labels.data_ptr(), p.stride, p.label_length.data_ptr(), p.logit_length.data_ptr(), 0,
      B, T, V, U, loss.data_ptr(), grad.data_ptr(), 0, grad.stride(0), grad.stride(1), None)


def old(i):
    assert lib.ctc_amd_loss_grad(*a1) == 0
    assert lib.ctc_amd_reduce_loss(loss.data_ptr(), B, out2.data_ptr(), st) == 0


def new(i):
    assert lib.ctc_amd_loss_grad_sum(*a2, sums[i % 3].data_ptr(), sums[(i + 1) % 3].data_ptr(), ws.data_ptr(), ws.numel(), st) == 0


for name, fn in (("warm", old), ("warm", new)):
    for i in range(20): fn(i)
torch.cuda.synchronize()
for rep in range(4):
    for name, fn in (("call + reduce launch", old), ("sum inside the launch", new)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): fn(i)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:24s} {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per step")
