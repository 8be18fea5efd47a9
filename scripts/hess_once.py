"""Diagnostic: one dense-Hessian call at config 5 (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = 32, 200, 32, 64
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
for _ in range(2):
    loss, _, h = ops.hessian(0, _lib.WRT_LOGITS, prep, want_grad=False)
torch.cuda.synchronize()
print(float(loss.sum()), float(h[0, 3, 5].abs().sum()))
