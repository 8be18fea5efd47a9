"""Diagnostic: loss + gradient per call of the three-kernel pipeline at a few wide-vocabulary shapes (CTC_AMD_LIB selects the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tf_seq2seq_losses_amd import _lib, ops
dev = torch.device("cuda:0")
print("library:", os.environ.get("CTC_AMD_LIB", "product"))
for kind in (0, 1):
    for (B, T, U, V) in ((32, 1000, 64, 4096), (32, 1000, 128, 4096), (64, 1000, 128, 2048), (32, 1000, 256, 2048)):
        host, d = bench.make_inputs(B, T, U, V, 0, False, dev)
        p = ops.Prepared(d["labels"], d["logits"], d["label_length"], d["logit_length"], 0, U=U)
        ws = ops._workspace(_lib.WS_LOSS_GRAD_LOGITS, kind, p)
        fn = lambda: ops.loss_grad(kind, 0, p, True, workspace=ws)
        bench.prewarm(fn, 40.0)
        ms, _ = bench._events_ms(fn, 40, 4)
        print(f"kind {kind} B {B} T {T} U {U} V {V} {ops.pipeline_of(kind, 0, p)}: {ms * 1e3:8.1f} us", flush=True)
