"""Diagnostic: Hessian-vector product at the north-star shape -- device time per call (HIP events around runs of 4 calls) of the
fused kernel + its (empty) fallback launches, of the log-domain pipeline, and of the fused kernel's timing modes (ctc_hvp_fused.hip
`mode`: 1 = phase 1 only, 2 = helpers alone in phase 2, 4 = chains alone in phase 2, 8 / 16 = phase 1 without the main chains'
steps / the E stage's arithmetic; results are meaningless in those)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
B, T, U, V = int(os.environ.get("F6_B", "256")), 1000, 128, 256
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
v = torch.randn((B, T, V), device="cuda:0")


def timed(kind, label):
    prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
    # the C call itself with everything allocated once (ops.hvp costs ~130 us of Python per call: the short timing modes would
    # measure that, not the kernel)
    lib = _lib.load()
    loss = torch.empty(B, dtype=torch.float32, device="cuda:0")
    out = torch.empty((B, T, V), dtype=torch.float32, device="cuda:0")
    ws = ops._workspace(_lib.WS_HVP, kind, prep)
    args = prep.common(kind, _lib.WRT_LOGITS) + (v.data_ptr(), loss.data_ptr(), None, out.data_ptr(), ws.data_ptr(), ws.numel())
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        rc = lib.ctc_amd_hvp(*args, st)
        assert rc == 0
    bench.prewarm(fn, 60.0)
    ms, _ = bench._events_ms(fn, 40, 4)
    print(f"{label:58s} {ms * 1e3:8.1f} us per call", flush=True)


for kind, kn in ((0, "classic"), (1, "simplified")):
    for mode, what in (("", "fused kernel + fallback launches"), ("v1", "log-domain pipeline (five launches)"), ("diag1", "fused, phase 1 only"),
                       ("diag2", "fused, phase 2 with the helpers alone"), ("diag4", "fused, phase 2 with the chains alone"),
                       ("diag9", "fused, phase 1 only, main chains idle"), ("diag17", "fused, phase 1 only, E stage loads only"),
                       ("diag25", "fused, phase 1 only, loads and barriers only")):
        _lib.debug_override("hvp", mode)
        timed(kind, f"{kn}: {what}")
    _lib.debug_override("hvp", "")
