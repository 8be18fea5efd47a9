"""Diagnostic: HVP kernel vs dense Hessian contraction vs float64 central differences (C oracle gradient)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from oracle import ctc_oracle as O
from tf_seq2seq_losses_amd import ops, _lib

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def fd64(kind, inp, v, eps):
    xp = (inp["logits"].astype(np.float64) + eps * v).astype(np.float32)
    xm = (inp["logits"].astype(np.float64) - eps * v).astype(np.float32)
    veff = (xp.astype(np.float64) - xm.astype(np.float64)) / (2 * eps)
    gp = C.loss_grad(kind, inp["labels"], xp, inp["label_length"], inp["logit_length"], 0)[1]
    gm = C.loss_grad(kind, inp["labels"], xm, inp["label_length"], inp["logit_length"], 0)[1]
    return (gp - gm) / (2 * eps), veff.astype(np.float32)


def run(kind, inp, dense, eps=1e-3):
    rng = np.random.default_rng(0)
    v = rng.standard_normal(inp["logits"].shape)
    fd, veff = fd64(kind, inp, v, eps)
    fd2, _ = fd64(kind, inp, v, eps * 2)
    fin = np.isfinite(fd).all(axis=(1, 2))
    p = ops.Prepared(t(inp["labels"]), t(inp["logits"]), t(inp["label_length"]), t(inp["logit_length"]), 0)
    out = ops.hvp(ops.KINDS[kind], _lib.WRT_LOGITS, p, t(veff))[2].cpu().numpy().astype(np.float64)
    sc = np.abs(fd[fin]).max()
    print(kind, inp["logits"].shape, "max|Hv|", sc, "fd(eps) vs fd(2eps)", np.abs(fd[fin] - fd2[fin]).max() / sc,
          "hvp vs fd", np.abs(out[fin] - fd[fin]).max() / sc)
    for b in range(len(fin)):
        if fin[b]:
            print("   b", b, np.abs(out[b] - fd[b]).max() / max(1e-30, np.abs(fd[b]).max()))
    if dense:
        hess = ops.hessian(ops.KINDS[kind], _lib.WRT_LOGITS, p, want_grad=False)[2]
        d = torch.einsum("btkuj,buj->btk", hess.double(), t(veff).double()).cpu().numpy()
        print("   dense vs fd", np.abs(d[fin] - fd[fin]).max() / sc, "dense vs hvp", np.abs(d[fin] - out[fin]).max() / sc)


for kind in ("classic", "simplified"):
    inp = O.generate_ctc_loss_inputs(4, 200, 130, 6, max_label_length=130)
    inp["labels"][0, :65] = 1
    run(kind, inp, True)
    B, T, U, V = 4, 1000, 128, 256
    rng = np.random.default_rng(11)
    inp = dict(logits=rng.standard_normal((B, T, V)).astype(np.float32), labels=rng.integers(1, V, (B, U)).astype(np.int32),
               label_length=np.array([128, 100, 64, 128], np.int32), logit_length=np.array([1000, 900, 1000, 517], np.int32))
    run(kind, inp, False)
