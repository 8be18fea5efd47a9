"""Diagnostic: a few loss+grad calls at the north-star config (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = 256, 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, B, T, V, U), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    loss, g = ops.loss_grad(0, _lib.WRT_LOGITS, prep, True, workspace=ws)
torch.cuda.synchronize()
print(float(loss.sum()))
