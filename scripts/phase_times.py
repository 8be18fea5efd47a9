"""Diagnostic: 40 x (loss-only call, gradient-resume call) at the north-star config; run under rocprofv3 --kernel-trace to get
the device time of phase 1 and phase 2 of the fused kernel separately (scripts/f6_phase_times.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = 256, 1000, int(os.environ.get("F6_U", "128")), int(os.environ.get("F6_V", "256"))
host, dev = bench.make_inputs(B, T, U, V, 2, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
kind = 1 if os.environ.get("F6_KINDNAME", "classic") == "simplified" else 0
for _ in range(40):
    loss, ws = ops.loss_forward(kind, _lib.WRT_LOGITS, prep)
    g = ops.grad_resume(kind, _lib.WRT_LOGITS, prep, ws)
torch.cuda.synchronize()
