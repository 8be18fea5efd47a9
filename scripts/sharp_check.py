"""Diagnostic: the north-star shape with sharp logits N(0, sigma^2): which flags the linear-domain kernel raises (histogram of the
per-utterance flag words), the time of the one-call and of the loss-only + resume form, and the error of every flagged utterance and
of a sample of the others against the float64 C oracle.   usage: [CTC_AMD_LIB=...] python scripts/sharp_check.py [sigma ...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
B, T, U, V = 256, 1000, 128, 256
print("library:", os.environ.get("CTC_AMD_LIB", "product"))
for sigma in [float(a) for a in sys.argv[1:]] or [1.0, 2.0, 3.0, 4.0]:
    host, d = bench.make_inputs(B, T, U, V, 0, False, dev, scale=sigma)
    p = ops.Prepared(d["labels"], d["logits"], d["label_length"], d["logit_length"], 0, U=U)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, B, T, V, U), dtype=torch.uint8, device=dev)

    def one():
        return ops.loss_grad(0, 0, p, True, workspace=ws)

    def two():
        l, w2 = ops.loss_forward(0, 0, p)
        return l, ops.grad_resume(0, 0, p, w2), w2
    res = {}
    for name, fn in (("one call", one), ("loss-only + resume", two)):
        for _ in range(20):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            out = fn()
        e1.record(); e1.synchronize()
        res[name] = (e0.elapsed_time(e1) / 40 * 1e3, out)
    fl1 = ops.fused_flags(ws, 0, p).cpu().numpy()
    fl2 = ops.fused_flags(res["loss-only + resume"][1][2], 0, p).cpu().numpy()
    idx = sorted(set(np.nonzero(fl1 | fl2)[0].tolist()) | set(range(4)))[:24]
    rl, rg = C.loss_grad("classic", host["labels"][idx], host["logits"][idx], host["label_length"][idx], host["logit_length"][idx], 0)
    g1 = res["one call"][1][1][idx].cpu().numpy(); g2 = res["loss-only + resume"][1][1][idx].cpu().numpy()
    l1 = res["one call"][1][0][idx].cpu().numpy(); l2 = res["loss-only + resume"][1][0][idx].cpu().numpy()
    e1 = np.abs(g1 - rg).max(axis=(1, 2)); e2 = np.abs(g2 - rg).max(axis=(1, 2))
    print(f"sigma {sigma:g}: one call {res['one call'][0]:6.1f} us, flags {dict(collections.Counter(hex(int(f)) for f in fl1))}; two calls {res['loss-only + resume'][0]:6.1f} us, flags {dict(collections.Counter(hex(int(f)) for f in fl2))}")
    print(f"     checked {len(idx)} utterances (all flagged + 4): worst gradient error one call {e1.max():.1e} (flagged {e1[fl1[idx] != 0].max() if (fl1[idx] != 0).any() else 0:.1e}), two calls {e2.max():.1e}; "
          f"loss rel. {np.abs(l1 - rl).max() / np.abs(rl).max():.1e} / {np.abs(l2 - rl).max() / np.abs(rl).max():.1e}", flush=True)
