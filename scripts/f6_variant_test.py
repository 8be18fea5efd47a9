"""Diagnostic for kernel experiments on the north-star instantiation (V = 256 contiguous float32, U <= 128): every utterance of a few
batches against the float64 C oracle -- one-call form and loss-only + gradient-resume form, N(0,1) and sharp logits, full and ragged
lengths -- with the number of utterances the linear-domain kernel flagged.  Works with -DCTC_F6_NS_ONLY variant libraries:
    CTC_AMD_LIB=scratch/libctc_v_<name>.so python scripts/f6_variant_test.py [classic|simplified]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops

dev = torch.device("cuda:0")
kind_name = sys.argv[1] if len(sys.argv) > 1 else "classic"
kind = ops.KINDS[kind_name]
print("library:", os.environ.get("CTC_AMD_LIB", "product"), kind_name, flush=True)
worst = 0.0
for (B, T, U, scale, ragged, seed) in ((32, 1000, 128, 1.0, False, 0), (32, 1000, 128, 3.0, False, 1), (48, 300, 128, 1.0, True, 2), (48, 140, 100, 3.0, True, 3),
                                       (16, 2000, 128, 1.0, False, 4), (64, 61, 24, 2.0, True, 5), (24, 997, 77, 1.0, True, 6), (8, 5000, 128, 1.0, False, 7)):
    rng = np.random.default_rng(seed)
    V = 256
    x = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    labels[0, : U // 2] = labels[0, 0]  # a run of repeats
    if ragged:
        tl = rng.integers(T // 2, T + 1, B).astype(np.int32); ll = rng.integers(U // 2, U + 1, B).astype(np.int32)
        tl[1], ll[1] = ll[1] + 3, ll[1]  # a nearly forced alignment
    else:
        tl = np.full(B, T, np.int32); ll = np.full(B, U, np.int32)
    rl, rg = C.loss_grad(kind_name, labels, x, ll, tl, 0)
    fin = np.isfinite(rl)
    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev), torch.from_numpy(tl).to(dev), 0, U=U)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, kind, B, T, V, U), dtype=torch.uint8, device=dev)
    loss, grad = ops.loss_grad(kind, 0, p, True, workspace=ws)
    torch.cuda.synchronize()
    fl = ops.fused_flags(ws, kind, p).cpu().numpy()
    ln, gn = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.array_equal(np.isfinite(ln), fin), "finite pattern"
    le = (np.abs(ln[fin] - rl[fin]) / np.maximum(1, np.abs(rl[fin]))).max() if fin.any() else 0.0
    ge = np.abs(gn - rg).max(axis=(1, 2))
    l2, ws2 = ops.loss_forward(kind, 0, p)
    g2 = ops.grad_resume(kind, 0, p, ws2)
    torch.cuda.synchronize()
    fl2 = ops.fused_flags(ws2, kind, p).cpu().numpy()
    l2n = l2.cpu().numpy()
    le2 = (np.abs(l2n[fin] - rl[fin]) / np.maximum(1, np.abs(rl[fin]))).max() if fin.any() else 0.0
    ge2 = np.abs(g2.cpu().numpy() - rg).max(axis=(1, 2))
    lin, red = fl == 0, fl != 0
    print(f"B={B:3d} T={T:5d} U={U:3d} scale={scale:g} {'ragged' if ragged else 'full  '}: one call loss {le:.1e} grad linear {ge[lin].max() if lin.any() else 0:.1e} redone {ge[red].max() if red.any() else 0:.1e} "
          f"flagged {int(red.sum())}/{B} (bits {hex(int(np.bitwise_or.reduce(fl)))}) | two calls loss {le2:.1e} grad {ge2.max():.1e} flagged {int((fl2 != 0).sum())}/{B} (bits {hex(int(np.bitwise_or.reduce(fl2)))})", flush=True)
    worst = max(worst, le, le2, ge.max(), ge2.max())
print("worst error", worst, "OK" if worst < 1e-4 else "ABOVE 1e-4")
