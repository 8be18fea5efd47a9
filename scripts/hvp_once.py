"""Diagnostic: a few Hessian-vector products at the north-star config (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = 256, 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
v = torch.randn((B, T, V), device="cuda:0")
for _ in range(3):
    loss, _, out = ops.hvp(0, _lib.WRT_LOGITS, prep, v)
torch.cuda.synchronize()
print(float(out.abs().max()))
