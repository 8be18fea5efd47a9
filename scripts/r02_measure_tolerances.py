"""Measures the quantities the relaxed test bounds are set from: run on the GPU box, prints one line each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def inputs(B, T, U, V, seed, ragged):
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((B, T, V), dtype=np.float32)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    if ragged:
        tl = rng.integers(T // 2, T, B, dtype=np.int32); ll = rng.integers(U // 2, U + 1, B, dtype=np.int32)
    else:
        tl = np.full(B, T, np.int32); ll = np.full(B, U, np.int32)
    return logits, labels, ll, tl


for kind in ("classic", "simplified"):
    for ragged, seed in ((False, 0), (True, 1)):
        B, T, U, V = 256, 1000, 128, 256
        logits, labels, ll, tl = inputs(B, T, U, V, seed, ragged)
        p = ops.Prepared(t(labels), t(logits), t(ll), t(tl), 0, U=U)
        for pipe in ("", "fused5"):
            _lib.debug_override("pipeline", pipe)
            loss, grad = ops.loss_grad(ops.KINDS[kind], 0, p, True)
            _lib.debug_override("pipeline", "")
            n = 8
            rl, rg = C.loss_grad(kind, labels[:n], logits[:n], ll[:n], tl[:n], 0)
            print(f"north-star {kind} {'ragged' if ragged else 'full'} pipeline {pipe or 'default(fused6)'}: max|dgrad| {np.abs(grad[:n].cpu().numpy() - rg).max():.3e} "
                  f"loss rel {(np.abs(loss[:n].cpu().numpy() - rl) / rl).max():.2e} row-sum {grad.sum(dim=2).abs().max().item():.2e}", flush=True)
    # T = 5000
    rng = np.random.default_rng(2)
    B, T, V, U = 3, 5000, 256, 128
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    labels = rng.integers(1, V, (B, U)).astype(np.int32)
    ll, tl = np.array([128, 77, 128], np.int32), np.array([5000, 4321, 2500], np.int32)
    p = ops.Prepared(t(labels), t(logits), t(ll), t(tl), 0, U=U)
    for pipe in ("", "fused5"):
        _lib.debug_override("pipeline", pipe)
        loss, grad = ops.loss_grad(ops.KINDS[kind], 0, p, True)
        _lib.debug_override("pipeline", "")
        rl, rg = C.loss_grad(kind, labels, logits, ll, tl, 0)
        print(f"T=5000 {kind} pipeline {pipe or 'default'}: per-utterance max|dgrad| {np.abs(grad.cpu().numpy() - rg).max(axis=(1, 2)).tolist()}", flush=True)
    # HVP at the north-star shape
    B, T, U, V = 4, 1000, 128, 256
    rng = np.random.default_rng(11)
    inp = dict(logits=rng.standard_normal((B, T, V)).astype(np.float32), labels=rng.integers(1, V, (B, U)).astype(np.int32),
               label_length=np.array([128, 100, 64, 128], np.int32), logit_length=np.array([1000, 900, 1000, 517], np.int32))
    v0 = rng.standard_normal((B, T, V))
    eps = 1e-3
    xp = (inp["logits"].astype(np.float64) + eps * v0).astype(np.float32); xm = (inp["logits"].astype(np.float64) - eps * v0).astype(np.float32)
    v = ((xp.astype(np.float64) - xm.astype(np.float64)) / (2 * eps)).astype(np.float32)
    gp = C.loss_grad(kind, inp["labels"], xp, inp["label_length"], inp["logit_length"], 0)[1]
    gm = C.loss_grad(kind, inp["labels"], xm, inp["label_length"], inp["logit_length"], 0)[1]
    fd = (gp - gm) / (2 * eps)
    pr = ops.Prepared(t(inp["labels"]), t(inp["logits"]), t(inp["label_length"]), t(inp["logit_length"]), 0)
    out = ops.hvp(ops.KINDS[kind], 0, pr, t(v))[2]
    outn = out.cpu().numpy().astype(np.float64)
    print(f"HVP {kind}: max|Hv - fd| / max|fd| per utterance {[float(np.abs(outn[b] - fd[b]).max() / np.abs(fd[b]).max()) for b in range(B)]}")
    u = rng.standard_normal((B, T, V)).astype(np.float32)
    hu = ops.hvp(ops.KINDS[kind], 0, pr, t(u))[2].double()
    a, b_ = (t(v).double() * hu).sum((1, 2)), (t(u).double() * out.double()).sum((1, 2))
    scale = t(v).double().flatten(1).norm(dim=1) * hu.flatten(1).norm(dim=1)
    print(f"HVP {kind}: |<v,Hu> - <u,Hv>| / (|v||Hu|) per utterance {((a - b_).abs() / scale).tolist()}; absolute {(a - b_).abs().tolist()} of {a.abs().tolist()}", flush=True)
