"""How sharp may logits be before a one-call loss + gradient starts to redo utterances in the log domain?  North-star shape, both
lattices: fraction of flagged utterances (posterior mass check D6 and friends) and the decay rate of the unnormalised P in bits per
frame, per sigma of N(0, sigma^2) logits.  The forward half's rate bounds (ctc_fused6.hip: RATE_MAX_X4_*) come from this table.
GPU tool; run from the repository root:  python tests/tools/rate_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, collections
import bench
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
B, T, U, V = 256, 1000, 128, 256
for kind in (0, 1):
    for sigma in (3.0, 3.25, 3.5, 3.75, 4.0, 4.5):
        host, d = bench.make_inputs(B, T, U, V, 0, False, dev, scale=sigma)
        p = ops.Prepared(d["labels"], d["logits"], d["label_length"], d["logit_length"], 0, U=U)
        ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, kind, B, T, V, U), dtype=torch.uint8, device=dev)
        loss, grad = ops.loss_grad(kind, 0, p, True, workspace=ws)
        fl = ops.fused_flags(ws, kind, p).cpu().numpy()
        rows = np.log2(np.exp(host["logits"][:8].astype(np.float64) - host["logits"][:8].max(axis=2, keepdims=True)).sum(axis=2)).sum(axis=1)
        rate = (loss[:8].cpu().numpy() / np.log(2) - rows) / T
        print("kind", kind, "sigma", sigma, "one-call flagged", (fl != 0).mean(), dict(collections.Counter(hex(int(f)) for f in fl)), "rate", np.round(rate[:3], 2), flush=True)
