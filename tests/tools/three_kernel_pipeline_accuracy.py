"""Gradient error of the three-kernel pipeline (the tier of vocabularies beyond 1024 and labels beyond 512) against the float64 C
oracle on sharp and long inputs.  GPU tool: python tests/tools/three_kernel_pipeline_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
print("gradient error of the three-kernel pipeline against the float64 C oracle")
for kind in ("classic", "simplified"):
    for (B, T, U, V, sigma) in ((8, 1000, 128, 2048, 1.0), (8, 1000, 128, 2048, 4.0), (8, 300, 100, 1500, 3.0), (4, 3000, 128, 2048, 1.0), (8, 1000, 600, 256, 3.0)):
        rng = np.random.default_rng(3)
        x = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
        labels = rng.integers(1, V, (B, U)).astype(np.int32)
        ll, tl = np.full(B, min(U, T // 2), np.int32), np.full(B, T, np.int32)
        rl, rg = C.loss_grad(kind, labels, x, ll, tl, 0)
        p = ops.Prepared(t(labels), t(x), t(ll), t(tl), 0, U=U)
        loss, grad = ops.loss_grad(ops.KINDS[kind], 0, p, True)
        print(f"{kind:10s} {ops.pipeline_of(ops.KINDS[kind], 0, p):8s} B {B:3d} T {T:5d} U {U:4d} V {V:4d} N(0,{sigma}^2): loss rel {np.abs(loss.cpu().numpy() - rl).max() / np.abs(rl).max():.1e}  grad {np.abs(grad.cpu().numpy() - rg).max():.1e}", flush=True)
