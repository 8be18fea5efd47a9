"""Diagnostic (test infrastructure): the linear-domain kernel on SHORT labels in LONG utterances (few labels = many forced blanks)
under a label bound that puts two label positions into a lane -- the regime of r04's soak failure (2 labels in 199 frames under a
bound of 100, N(0, 3^2): the state that carries the posterior sits 2^-105 below a lane-mate).  Per cell: fraction of utterances the
one-call form flags (D6: the sound detector) | fraction the forward half flags, and the forward losses that are off by >= 1e-4.
usage: python tests/tools/flag_stats_short_labels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(17)
B = 256
Ts = (32, 64, 128, 256, 512)
print("sigma   V bound  ll | " + " ".join(f"T={t:<22d}" for t in Ts))
for sigma in (2.0, 3.0):
    for V in (29, 256):
        for bound in (64, 128):
            for llv in (1, 2, 4, 8, 16):
                cells = []
                for T in Ts:
                    labels = rng.integers(1, V, (B, bound), dtype=np.int32)
                    ll = np.full(B, llv, np.int32); tl = np.full(B, T, np.int32)
                    x = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
                    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev), torch.from_numpy(tl).to(dev), 0, U=bound)
                    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, B, T, V, bound), dtype=torch.uint8, device=dev)
                    loss, grad = ops.loss_grad(0, 0, p, True, workspace=ws)
                    fl = ops.fused_flags(ws, 0, p).cpu().numpy()
                    l1, ws1 = ops.loss_forward(0, 0, p)
                    fl1 = ops.fused_flags(ws1, 0, p).cpu().numpy()
                    rl, rg = C.loss_grad("classic", labels, x, ll, tl, 0)
                    le = np.abs(l1.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))
                    ge = np.abs(grad.cpu().numpy() - rg).max(axis=(1, 2))
                    cells.append(f"{(fl != 0).mean():5.3f}|{(fl1 != 0).mean():4.2f} bad {int((le >= 1e-4).sum()):3d} g{ge.max():.0e}")
                print(f"{sigma:5.1f} {V:4d} {bound:5d} {llv:3d} | " + " ".join(cells), flush=True)
