"""Soak run (test infrastructure): random shapes through ctc_amd_loss_grad for a bounded time, each checked against the
float64 C oracle on a subsample; prints a progress line every few seconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(12345)
dev = torch.device("cuda:0")
t0 = time.time(); n = 0; worst = 0.0; last = t0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 400)); T = int(rng.integers(1, 300)); V = int(rng.choice([3, 8, 29, 64, 256, 300, 512, 1000, 1500, 2048, 4100]))
    U = int(rng.choice([0, 1, 7, 40, 64, 100, 128, 200, 256, 300, 400, 512, 600]))
    if V > 1500: B = min(B, 40)
    kind = int(rng.integers(0, 2))
    x = rng.standard_normal((B, T, V)).astype(np.float32) * float(rng.choice([0.3, 1.0, 3.0]))
    labels = rng.integers(1, max(V, 2), (B, max(U, 1))).astype(np.int32) % V
    ll = rng.integers(0, U + 1, B).astype(np.int32); tl = rng.integers(0, T + 1, B).astype(np.int32)
    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev),
                     torch.from_numpy(tl).to(dev), 0, U=max(U, 1))
    mode = int(rng.integers(0, 3))  # one call / loss-only call + gradient-resume call / one call with the in-launch loss sum
    if mode == 0:
        loss, grad = ops.loss_grad(kind, _lib.WRT_LOGITS, p, True)
    elif mode == 1:
        loss, ws = ops.loss_forward(kind, _lib.WRT_LOGITS, p)
        grad = ops.grad_resume(kind, _lib.WRT_LOGITS, p, loss, ws)
    else:
        sum2 = torch.zeros(2, dtype=torch.int64, device=dev)
        loss, grad = ops.loss_grad_sum(kind, _lib.WRT_LOGITS, p, sum2)
        fin_t = torch.isfinite(loss)
        assert int(sum2[1]) == int(fin_t.sum()) and int(sum2[0]) == int(torch.round(loss[fin_t].double() * 1048576.0).sum()), (B, T, V, U, kind)
    m = min(B, 6)
    rl, rg = C.loss_grad("classic" if kind == 0 else "simplified", labels[:m], x[:m], ll[:m], tl[:m], 0)
    ln, gn = loss[:m].cpu().numpy(), grad[:m].cpu().numpy()
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(ln), fin), (B, T, V, U, kind)
    err = float(np.abs(gn - rg).max())
    big = fin.any() and np.abs(rl[fin]).max() > 500
    assert err < (2e-3 if big else 2e-4), (B, T, V, U, kind, err)
    assert torch.isfinite(grad).all()
    worst = max(worst, err); n += 1
    if time.time() - last > 5:
        print(f"{n} cases, worst gradient error {worst:.2e}", flush=True); last = time.time()
print(f"soak ok: {n} random cases in {time.time() - t0:.0f} s, worst gradient error {worst:.2e}")
