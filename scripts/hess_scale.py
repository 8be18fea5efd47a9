"""Diagnostic: dense-Hessian kernel time vs batch size (latency of one slab sweep vs throughput)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
T, U, V = 200, 32, 64
for B in (1, 2, 4, 8, 16, 32):
    host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
    prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
    for _ in range(2):
        ops.hessian(0, _lib.WRT_LOGITS, prep, want_grad=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ws = torch.empty(_lib.workspace_bytes(_lib.WS_HESSIAN, 0, B, T, V, U), dtype=torch.uint8, device="cuda:0")
    loss = torch.empty(B, device="cuda:0"); hess = torch.empty((B, T, V, T, V), device="cuda:0")
    lib = _lib.load()
    def call():
        rc = lib.ctc_amd_hessian(*prep.common(0, _lib.WRT_LOGITS), loss.data_ptr(), None, hess.data_ptr(), ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    call(); torch.cuda.synchronize()
    e0.record()
    for _ in range(3): call()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: {e0.elapsed_time(e1)/3:.3f} ms per call  ({B*(T*V)**2*4/1e9:.2f} GB)", flush=True)
    del hess
