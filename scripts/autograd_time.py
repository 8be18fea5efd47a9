"""Diagnostic: user-visible time of loss + backward through the autograd wrapper at the north-star shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import tf_seq2seq_losses_amd as ctc
B, T, U, V = 256, 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
x = dev["logits"].requires_grad_(True)


def step():
    loss = ctc.classic_ctc_loss(dev["labels"], x, dev["label_length"], dev["logit_length"], 0)
    (g,) = torch.autograd.grad(loss.sum(), x)
    return g


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
t_submit = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host submit time per step {t_submit / 100 * 1e3:.3f} ms")
print(f"classic_ctc_loss + autograd.grad(loss.sum()) at B={B} T={T} U={U} V={V}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms")
