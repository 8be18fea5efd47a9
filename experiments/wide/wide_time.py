"""Diagnostic: the EXPERIMENTAL one-launch wide-vocabulary tier (csrc/ctc_wide.hip, not part of the product library: build the
diagnostic library with scripts/build_wide_variant.sh and run with CTC_AMD_LIB=scratch/libctc_wide_diag.so) against the three-kernel
pipeline at several shapes; HIP events around runs of calls; prints a markdown table (profiles/r03_wide_time.md).
usage: python scripts/wide_time.py [B,T,U,V ...] [diagN ...]   (diagN: timing diagnostics of the wide tier, bit set described in
ctc_wide.hip; they exist in CTC_DIAG builds only: scripts/build_wide_variant.sh, then CTC_AMD_LIB=scratch/libctc_wide_diag.so)"""
import os, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops

diags = [a for a in sys.argv[1:] if a.startswith("diag")]
sys.argv = [a for a in sys.argv if not a.startswith("diag")]
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(32, 1000, 128, 4096), (64, 1000, 128, 2048), (16, 1000, 128, 8192),
                                                                          (128, 1000, 128, 4096), (32, 500, 64, 4096), (8, 1000, 128, 16384)]
dev = torch.device("cuda:0")
print("| B | T | U | V | pipeline | us per call (median / min) | algorithmic GB/s | fraction of 8 TB/s |")
print("|--:|--:|--:|--:|:--|--:|--:|--:|")
for B, T, U, V in shapes:
    host, d = bench.make_inputs(B, T, U, V, 2, False, dev)
    p = ops.Prepared(d["labels"], d["logits"], d["label_length"], d["logit_length"], 0, U=U)
    ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD_LOGITS, 0, B, T, V, U), dtype=torch.uint8, device=dev)
    for pl in ["wide", "v1"] + diags:
        _lib.debug_override("pipeline", "wide" if pl.startswith("diag") else pl)
        _lib.debug_override("wide", pl if pl.startswith("diag") else "")
        name = pl
        for _ in range(10):
            ops.loss_grad(0, _lib.WRT_LOGITS, p, True, workspace=ws)
        torch.cuda.synchronize()
        ts = []
        for r in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.loss_grad(0, _lib.WRT_LOGITS, p, True, workspace=ws)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 4)
        _lib.debug_override("pipeline", "")
        _lib.debug_override("wide", "")
        med = st.median(ts)
        gbs = 2.0 * B * T * V * 4 / med / 1e3
        print(f"| {B} | {T} | {U} | {V} | {name} | {med:.1f} / {min(ts):.1f} | {gbs:.0f} | {gbs / 8000:.3f} |", flush=True)
