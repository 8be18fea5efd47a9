"""Reads the row-loop cycle stamps of a -DCTC_HESS_STAMPS build (CTC_AMD_LIB=scratch/libctc_hst.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib, ops
T, U, V = 200, 32, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
host, dev = bench.make_inputs(B, T, U, V, 0, False, torch.device("cuda:0"))
prep = ops.Prepared(dev["labels"], dev["logits"], dev["label_length"], dev["logit_length"], 0, U=U)
lib = _lib.load()
ws = torch.zeros(_lib.workspace_bytes(_lib.WS_HESSIAN, 0, B, T, V, U), dtype=torch.uint8, device="cuda:0")
loss = torch.empty(B, device="cuda:0"); hess = torch.empty((B, T, V, T, V), device="cuda:0")
for _ in range(2):
    rc = lib.ctc_amd_hessian(*prep.common(0, _lib.WRT_LOGITS), loss.data_ptr(), None, hess.data_ptr(), ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
al = lambda x: (x + 255) & ~255
UP, ERS, SRS = 64, 68, 2 * 64 + 8
o = 0; o = al(o + B * T * ERS * 4); o = al(o + B * (T + 1) * SRS * 4); o = al(o + B * (T + 1) * SRS * 4); o = al(o + B * 8)
st = ws[o:o + 64].view(torch.int64).cpu().numpy()
n = max(1, int(st[6]))
names = ["loop top (vmcnt wait of the prefetch)", "issue prefetch + scale", "lattice step + sums", "emit_row after half_sum",
         "renorm + flush", "emit_row up to half_sum"]
for i, nm in enumerate(names):
    print(f"{nm:42s} {st[i] / n:8.0f} cycles/row")
print("rows", n, "total/row", sum(st[:6]) / n)
