"""Diagnostic: per-wavefront work / barrier-wait cycles of the linear-domain fused kernel, from a -DCTC_F6_STAMPS build
(scripts/build_f6_variant.sh -DCTC_F6_STAMPS; CTC_AMD_LIB=scratch/libctc_f6v.so python scripts/f6_stamps.py).
Two s_memtime reads per block barrier and wavefront; the role that waits least at the barriers is the one the others wait for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = int(os.environ.get("F6_B", "256")), 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
sel = _lib.WS_LOSS_GRAD_LOGITS
ws = torch.zeros(_lib.workspace_bytes(sel, 0, B, T, V, U), dtype=torch.uint8, device="cuda:0")
for _ in range(20):
    ops.loss_grad(0, _lib.WRT_LOGITS, prep, True, workspace=ws)
torch.cuda.synchronize()
off = _lib.flags_offset(0, B, T, V, U) + 4 * B
NW = 12
st = ws[off:off + B * 16 * 32].view(torch.int64).cpu().numpy().reshape(B, 16, 4)[:, :NW]  # (b * 16 + wave) * 4 words
names = ["main A", "main B", "recompute A", "recompute B"] + [f"helper A{h}" for h in range(4)] + [f"helper B{h}" for h in range(4)]
print(f"B={B}: mean over the utterances, in us at 2.36 GHz (s_memtime counts shader clocks; the stamps themselves slow the kernel by ~20 %)")
for i, nm in enumerate(names):
    w1, q1, w2, q2 = (st[:, i, k].mean() / 2360.0 for k in range(4))
    print(f"{nm:12s}: phase 1 work {w1:6.1f} us wait {q1:6.1f} | phase 2 work {w2:6.1f} wait {q2:6.1f} | total {w1 + q1 + w2 + q2:6.1f}")

