"""Gradient error of the log-domain fused tier alone (pipeline forced to fused5: the roles that redo what the linear-domain kernel
flags) against the float64 C oracle, at the shapes DESIGN.md section 2 quotes.  GPU tool: python tests/tools/log_domain_tier_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
print("gradient error of the log-domain fused tier (pipeline forced to fused5) against the float64 C oracle")
for kind in ("classic", "simplified"):
    for (B, T, U, V, sigma) in ((16, 1000, 128, 256, 1.0), (16, 1000, 128, 256, 4.0), (3, 5000, 128, 256, 1.0), (8, 1000, 512, 256, 1.0), (8, 600, 40, 64, 6.0)):
        rng = np.random.default_rng(2)
        x = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
        labels = rng.integers(1, V, (B, U)).astype(np.int32)
        ll, tl = np.full(B, U, np.int32), np.full(B, T, np.int32)
        rl, rg = C.loss_grad(kind, labels, x, ll, tl, 0)
        _lib.debug_override("pipeline", "fused5")
        try:
            p = ops.Prepared(t(labels), t(x), t(ll), t(tl), 0, U=U)
            loss, grad = ops.loss_grad(ops.KINDS[kind], 0, p, True)
        finally:
            _lib.debug_override("pipeline", "")
        print(f"{kind:10s} B {B:3d} T {T:5d} U {U:4d} V {V:4d} N(0,{sigma}^2): loss rel {np.abs(loss.cpu().numpy() - rl).max() / np.abs(rl).max():.1e}  grad {np.abs(grad.cpu().numpy() - rg).max():.1e}", flush=True)
