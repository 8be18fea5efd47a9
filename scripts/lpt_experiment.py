"""Diagnostic: does longest-first ordering of a ragged batch help when B exceeds the CU count?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
for B in (512, 1024, 2048):
    host, dev = bench.make_inputs(B, 1000, 128, 256, 1, True, torch.device("cuda:0"))
    order = np.argsort(-host["logit_length"], kind="stable")
    for name, idx in (("as given", np.arange(B)), ("longest first", order)):
        ix = torch.from_numpy(idx).to("cuda:0")
        prep = ops.Prepared(dev["labels"][ix].contiguous(), dev["logits"][ix].contiguous(), dev["label_length"][ix].contiguous(),
                            dev["logit_length"][ix].contiguous(), 0, U=128)
        ws = torch.empty(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, 0, B, 1000, 256, 128), dtype=torch.uint8, device="cuda:0")
        for _ in range(5): ops.loss_grad(0, 0, prep, True, workspace=ws)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ops.loss_grad(0, 0, prep, True, workspace=ws)
        torch.cuda.synchronize()
        print(f"B={B} {name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms", flush=True)
        del prep, ws
