import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ctc_oracle as O
from tf_seq2seq_losses_amd import _lib, ops
rng = np.random.default_rng(int(sys.argv[7]) if len(sys.argv) > 7 else 1)
dev = torch.device("cuda:0")
B, T, V, U, kind, blank = (int(a) for a in sys.argv[1:7])
kn = "classic" if kind == 0 else "simplified"
sc = float(sys.argv[8]) if len(sys.argv) > 8 else 1.0
x = (rng.standard_normal((B, T, V)) * sc).astype(np.float32)
tok = np.array([k for k in range(V) if k != blank])
labels = tok[rng.integers(0, V - 1, (B, U))].astype(np.int32)
ll = rng.integers(0, U + 1, B).astype(np.int32); tl = rng.integers(0, T + 1, B).astype(np.int32)
v = rng.standard_normal((B, T, V)).astype(np.float32)
print(B, T, V, U, kn, blank, sc)
p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev), torch.from_numpy(tl).to(dev), blank, U=U)
vt = torch.from_numpy(v).to(dev)
loss, _, out = ops.hvp(kind, _lib.WRT_LOGITS, p, vt)
_lib.debug_override("hvp", "v1")
loss1, _, out1 = ops.hvp(kind, _lib.WRT_LOGITS, p, vt)
_lib.debug_override("hvp", "")
def grad(z):
    d = O.ctc_loss(kn, labels, z, ll, tl, blank)
    g = O.logits_gradient(d, z)
    return np.where(np.isfinite(d.loss)[:, None, None], g, 0.0)
x64, v64, eps = x.astype(np.float64), v.astype(np.float64), 2e-3
d1 = (grad(x64 + eps * v64) - grad(x64 - eps * v64)) / (2 * eps)
d2 = (grad(x64 + 2 * eps * v64) - grad(x64 - 2 * eps * v64)) / (4 * eps)
fd = (4.0 * d1 - d2) / 3.0
o, o1 = out.cpu().numpy(), out1.cpu().numpy()
for b in range(B):
    m = max(1e-9, np.abs(fd[b]).max())
    ef, e1 = np.abs(o[b] - fd[b]).max(), np.abs(o1[b] - fd[b]).max()
    if ef > 1e-5 or e1 > 1e-5:
        t, k = np.unravel_index(np.argmax(np.abs(o[b] - fd[b])), fd[b].shape)
        print(f"b={b} ll={ll[b]} tl={tl[b]} max|fd|={m:.3e} fused err {ef:.3e} v1 err {e1:.3e} loss {float(loss[b]):.4f}/{float(loss1[b]):.4f} worst at t={t} k={k}: fused {o[b,t,k]:.5f} v1 {o1[b,t,k]:.5f} fd {fd[b,t,k]:.5f}")
print("max fused err", np.abs(o - fd).max(), "max v1 err", np.abs(o1 - fd).max())
