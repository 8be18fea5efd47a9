"""Diagnostic (test infrastructure: uses the oracle): replays a case dumped by tests/tools/soak.py through every pipeline and
prints, per utterance, loss and gradient error against the float64 C oracle.  usage: python tests/tools/replay_case.py case.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
d = np.load(sys.argv[1], allow_pickle=True)
x, labels, ll, tl, kind, U = d["x"], d["labels"], d["ll"], d["tl"], int(d["kind"]), int(d["U"])
print("case:", str(d["why"]), "shape", x.shape, "U", U, "kind", kind)
dev = torch.device("cuda:0")
m = min(x.shape[0], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
rl, rg = C.loss_grad("classic" if kind == 0 else "simplified", labels[:m], x[:m], ll[:m], tl[:m], 0)
p = ops.Prepared(*(torch.from_numpy(a).to(dev) for a in (labels, x, ll, tl)), 0, U=U)
for pl in ("", "fused5", "v1", "wide"):
    _lib.debug_override("pipeline", pl)
    name = ops.pipeline_of(kind, _lib.WRT_LOGITS, p)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, kind, *x.shape, U), dtype=torch.uint8, device=dev)
    loss, grad = ops.loss_grad(kind, _lib.WRT_LOGITS, p, True, workspace=ws)
    flags = ops.fused_flags(ws, kind, p)[:m].cpu().numpy() if name == "fused6" else None
    _lib.debug_override("pipeline", "")
    ln, gn = loss[:m].cpu().numpy(), grad[:m].cpu().numpy()
    print(f"--- override {pl!r} -> pipeline {name}")
    for b in range(m):
        ge = float(np.abs(gn[b] - rg[b]).max()) if gn[b].size else 0.0
        print(f"  b={b} ll={ll[b]} tl={tl[b]} flag={None if flags is None else hex(int(flags[b]))} loss {ln[b]:.6f} ref {rl[b]:.6f} "
              f"rel {abs(ln[b] - rl[b]) / max(1.0, abs(rl[b])) if np.isfinite(rl[b]) else 0:.2e} grad err {ge:.2e}")
        if ge > 1e-3:
            bad = np.nonzero(np.abs(gn[b] - rg[b]).max(axis=1) > 1e-3)[0]
            print(f"     frames off by more than 1e-3: {bad.tolist()[:60]} ({len(bad)} of {tl[b]})")
