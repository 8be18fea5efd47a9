"""Diagnostic: host-side profile (cProfile) of classic_ctc_loss + autograd.grad at the north-star shape."""
import cProfile, pstats, os, sys, time
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


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host submit {t1 / 200 * 1e6:.1f} us per step; with sync {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
