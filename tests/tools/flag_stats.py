"""Diagnostic (test infrastructure): how often does the linear-domain kernel flag an utterance, as a function of how BINDING the
alignment is (slack = frames - labels - repeats) and how sharp the logits are?  One loss+gradient call per cell (the posterior mass
check D6 is the sound detector of lost mass), flags read back from the workspace; every flagged and a sample of unflagged utterances
can be checked against the C oracle with --check.
usage: python tests/tools/flag_stats.py [--check] [classic|simplified]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
kind_name = "simplified" if "simplified" in sys.argv else "classic"
kind = ops.KINDS[kind_name]
CHECK = "--check" in sys.argv
rng = np.random.default_rng(7)
B = 256
print(f"{kind_name}: fraction of {B} utterances flagged by the one-call form (bits seen) | by the loss-only call; rows: sigma, V, U; columns: slack")
slacks = (0, 1, 2, 4, 8, 16, 32, 64, 128, 512)
print("sigma    V    U | " + " ".join(f"{s:>18d}" for s in slacks))
for sigma in (1.0, 3.0, 5.0):
    for V in (3, 8, 64, 256):
        for U in (8, 32, 128):
            cells = []
            bad_g, bad_r, bad_l = 0.0, 0.0, []
            for slack in slacks:
                labels = rng.integers(1, V, (B, U), dtype=np.int32)
                ll = np.full(B, U, np.int32)
                rep = (labels[:, 1:] == labels[:, :-1]).sum(axis=1) if kind == 0 else np.zeros(B, np.int64)
                tl = (U + rep + slack).astype(np.int32)
                T = int(tl.max())
                x = (rng.standard_normal((B, T, V)) * sigma).astype(np.float32)
                p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(ll).to(dev), torch.from_numpy(tl).to(dev), 0, U=U)
                if ops.pipeline_of(kind, _lib.WRT_LOGITS, p) != "fused6":
                    cells.append("      -      "); continue
                ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, kind, B, T, V, U), dtype=torch.uint8, device=dev)
                loss, grad = ops.loss_grad(kind, 0, p, True, workspace=ws)
                fl = ops.fused_flags(ws, kind, p).cpu().numpy()
                l1, ws1 = ops.loss_forward(kind, 0, p)
                fl1 = ops.fused_flags(ws1, kind, p).cpu().numpy()
                cells.append(f"{(fl != 0).mean():5.3f}({int(np.bitwise_or.reduce(fl)):3x})|{(fl1 != 0).mean():4.2f}({int(np.bitwise_or.reduce(fl1)):3x})")
                if CHECK:  # every utterance against the float64 C oracle: the worst gradient error (one call) and what the loss-only call gets wrong
                    from oracle import c_oracle as C
                    rl, rg = C.loss_grad(kind_name, labels, x, ll, tl, 0)
                    err = np.abs(grad.cpu().numpy() - rg).max(axis=(1, 2))
                    le = np.abs(l1.cpu().numpy() - rl) / np.maximum(1, np.abs(rl))
                    bad_g = max(bad_g, float(err[fl == 0].max()) if (fl == 0).any() else 0.0)
                    bad_r = max(bad_r, float(err[fl != 0].max()) if (fl != 0).any() else 0.0)
                    nbad = int((le >= 1e-4).sum())
                    if nbad:
                        bad_l.append((slack, nbad, f"{le.max():.1e}", "unflagged:" + str(int(((le >= 1e-4) & (fl1 == 0)).sum()))))
            print(f"{sigma:5.1f} {V:4d} {U:4d} | " + " ".join(cells) + (f"  || worst grad err linear {bad_g:.1e} redone {bad_r:.1e}; loss-only errors >= 1e-4 (slack, count, worst, of them unflagged): {bad_l}" if CHECK else ""), flush=True)
