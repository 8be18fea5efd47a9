"""Soak run of the Hessian-vector product (test infrastructure): random shapes inside the fused kernel's range through
ctc_amd_hvp, the first utterances of each compared with the Richardson-extrapolated directional derivative of the float64
NumPy oracle's gradient -- 1e-4 of max|Hv| for utterances the fused kernel kept in the linear domain (its flag word) and, since the log-domain rows take
posteriors and tangents relative to the frame's own mass, also for
those it redid in the log domain (sharp logits: the float32 log-domain recursion's own accuracy) -- and with the log-domain
pipeline forced through the override (2e-4).  usage: python tests/tools/soak_hvp.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ctc_oracle as O
from tf_seq2seq_losses_amd import _lib, ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
dev = torch.device("cuda:0")
t0 = time.time(); n = 0; last = t0; worst_pair = 0.0
worst = {"linear": 0.0, "redone": 0.0}; count = {"linear": 0, "redone": 0}
while time.time() - t0 < budget:
    B = int(rng.integers(1, 40)); T = int(rng.integers(1, 300)); V = int(rng.choice([4, 8, 28, 64, 128, 252, 256]))
    U = int(rng.choice([1, 2, 7, 40, 64, 65, 100, 128]))
    kind = int(rng.integers(0, 2)); kn = "classic" if kind == 0 else "simplified"
    blank = int(rng.integers(0, V))
    x = (rng.standard_normal((B, T, V)) * float(rng.choice([0.3, 1.0, 2.0]))).astype(np.float32)
    tok = np.array([k for k in range(V) if k != blank])
    labels = tok[rng.integers(0, V - 1, (B, U))].astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32); tl = rng.integers(0, T + 1, B).astype(np.int32)
    v = rng.standard_normal((B, T, V)).astype(np.float32)
    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev),
                     torch.from_numpy(tl).to(dev), blank, U=U)
    vt = torch.from_numpy(v).to(dev)
    loss, _, out, ws = ops.hvp(kind, _lib.WRT_LOGITS, p, vt, return_workspace=True)
    off = _lib.hvp_flags_offset(kind, B, T, V, U)
    flags = ws[off:off + 4 * B].view(torch.int32).cpu().numpy() if T > 0 else np.zeros(B, np.int32)
    _lib.debug_override("hvp", "v1")
    try:
        loss1, _, out1 = ops.hvp(kind, _lib.WRT_LOGITS, p, vt)
    finally:
        _lib.debug_override("hvp", "")
    assert torch.isfinite(out).all(), (B, T, V, U, kind, blank)
    assert torch.equal(torch.isfinite(loss), torch.isfinite(loss1)), (B, T, V, U, kind, blank)
    scale = max(1.0, float(out1.abs().max()))
    err = float((out - out1).abs().max()) / scale
    assert err < 2e-4, (B, T, V, U, kind, blank, err)  # (both within 1e-4 of the float64 derivative)
    worst_pair = max(worst_pair, err)
    if B * T * V < 400000 and T > 0:   # float64 check of the first utterances
        m = min(B, 3)
        fin = np.isfinite(loss1[:m].cpu().numpy())

        def grad(z):
            d = O.ctc_loss(kn, labels[:m], z, ll[:m], tl[:m], blank)
            return np.where(fin[:, None, None], O.logits_gradient(d, z), 0.0)
        x64, v64, eps = x[:m].astype(np.float64), v[:m].astype(np.float64), 2e-3
        d1 = (grad(x64 + eps * v64) - grad(x64 - eps * v64)) / (2 * eps)
        d2 = (grad(x64 + 2 * eps * v64) - grad(x64 - 2 * eps * v64)) / (4 * eps)
        fd = (4.0 * d1 - d2) / 3.0
        on = out[:m].cpu().numpy()
        for b in range(m):
            cls = "linear" if flags[b] == 0 else "redone"
            e = float(np.abs(on[b] - fd[b]).max()) / max(1.0, float(np.abs(fd[b]).max()))
            assert e < 1e-4, (cls, B, T, V, U, kind, blank, b, e, int(flags[b]))  # (r02 / early r03: 2e-3 for redone utterances)
            worst[cls] = max(worst[cls], e); count[cls] += 1
    n += 1
    if time.time() - last > 5:
        print(f"{n} cases; vs float64: " + ", ".join(f"{k} {worst[k]:.2e} ({count[k]})" for k in worst) + f"; vs log-domain pipeline {worst_pair:.2e}", flush=True); last = time.time()
print(f"hvp soak ok: {n} random cases in {time.time() - t0:.0f} s; worst error / max(1, max|Hv|) against the float64 directional derivative: "
      + ", ".join(f"{k} {worst[k]:.2e} ({count[k]} utterances, bound 1e-4)" for k in worst)
      + f"; against the log-domain pipeline {worst_pair:.2e} (bound 2e-4)")
