"""Diagnostic for the stale-row defect of the experimental wide tier: one batch through the three kernels and, six
times, through the wide tier of the diagnostic library; prints call time, the number of gradient rows that differ and where.
    bash scripts/build_wide_variant.sh && CTC_AMD_LIB=scratch/libctc_wide_diag.so python tests/tools/wide_race.py [B T V U [scale]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
B, T, V, U = [int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (270, 492, 1028, 200))]
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
rng = np.random.default_rng(5)
x = torch.from_numpy((rng.standard_normal((B, T, V)) * scale).astype(np.float32)).to(dev)
labels = torch.from_numpy(rng.integers(1, V, (B, U)).astype(np.int32)).to(dev)
ll = torch.from_numpy(rng.integers(0, U + 1, B).astype(np.int32)).to(dev); tl = torch.from_numpy(rng.integers(0, T + 1, B).astype(np.int32)).to(dev)
p = ops.Prepared(labels, x, ll, tl, 0, U=U)
_lib.debug_override("pipeline", "v1")
l1, g1 = ops.loss_grad(0, 0, p, True)
_lib.debug_override("pipeline", "wide")
assert ops.pipeline_of(0, 0, p) == "wide"
nbytes = _lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, B, T, V, U)
L_off = None
for rep in range(6):
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    l2, g2 = ops.loss_grad(0, 0, p, True, workspace=ws)
    torch.cuda.synchronize(); dt = time.time() - t0
    bad = ((g2 - g1).abs().amax(dim=2) > 1e-3)
    nb = int(bad.sum())
    where = [(int(b), int(t)) for b, t in bad.nonzero()[:6].tolist()]
    print(f"rep {rep}: {dt * 1e3:.2f} ms, rows off by > 1e-3: {nb} {where}, losses equal: {bool(torch.equal(l1, l2))}", flush=True)
