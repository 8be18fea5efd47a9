"""Diagnostic: dense Hessian at shapes that take the general (one slab per wavefront) kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
for (B, T, U, V) in ((8, 200, 64, 64), (4, 300, 100, 64), (2, 150, 40, 256)):
    host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
    prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
    for _ in range(2):
        ops.hessian(0, _lib.WRT_LOGITS, prep, want_grad=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        h = ops.hessian(0, _lib.WRT_LOGITS, prep, want_grad=False)[2]
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    gb = B * (T * V) ** 2 * 4 / 1e9
    print(f"B={B} T={T} U={U} V={V}: {ms:.3f} ms, {gb:.2f} GB -> {gb / ms:.2f} TB/s", flush=True)
    del h
