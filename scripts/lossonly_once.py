"""Diagnostic: a few loss-only calls (phase 1 of the fused kernel) and a few gradient-resume calls (phase 2) at the north-star config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = 256, 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
for _ in range(3):
    loss, ws = ops.loss_forward(0, _lib.WRT_LOGITS, prep)
    g = ops.grad_resume(0, _lib.WRT_LOGITS, prep, ws)
torch.cuda.synchronize()
print(float(loss.sum()))
