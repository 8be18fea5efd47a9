"""Diagnostic: phase 1 alone (loss-only call), phase 2 alone (gradient-resume call) and the one-launch call of several library builds,
interleaved in one process at the north-star shape (HIP events around runs of 8 calls).
usage: python scripts/phase_ab.py tree pk p1 ...   (names as scripts/ab_time.py)"""
import ctypes, os, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, U, V = int(os.environ.get("F6_B", "256")), 1000, 128, 256
kind = 1 if os.environ.get("F6_KINDNAME", "classic") == "simplified" else 0
names = sys.argv[1:] or ["tree"]
libs = []
for n in names:
    path = os.path.join(ROOT, "tf_seq2seq_losses_amd", "libctc_amd.so") if n == "tree" else os.path.join(ROOT, "scratch", f"libctc_v_{n}.so")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = restype; fn.argtypes = argtypes
    libs.append(lib)
dev = torch.device("cuda:0")
host, d = bench.make_inputs(B, T, U, V, 2, False, dev)
need = 0
nbytes = ctypes.c_size_t()
for lib in libs:
    assert lib.ctc_amd_workspace_bytes(_lib.WS_LOSS_GRAD, kind, B, T, V, U, ctypes.byref(nbytes)) == 0
    need = max(need, nbytes.value)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
loss = torch.empty(B, device=dev); grad = torch.empty(B, T, V, device=dev)
stream = torch.cuda.current_stream().cuda_stream
x = d["logits"]


def call(lib, want_grad=True):
    rc = lib.ctc_amd_loss_grad(kind, 0, x.data_ptr(), d["labels"].data_ptr(), d["labels"].shape[1], d["label_length"].data_ptr(),
                               d["logit_length"].data_ptr(), 0, B, T, V, U, loss.data_ptr(), grad.data_ptr() if want_grad else None, None, ws.data_ptr(), need, stream)
    assert rc == 0, lib.ctc_amd_last_error()


def resume(lib):
    rc = lib.ctc_amd_grad_resume(kind, 0, x.data_ptr(), _lib.F32, x.stride(0), x.stride(1), d["labels"].data_ptr(), d["labels"].shape[1],
                                 d["label_length"].data_ptr(), d["logit_length"].data_ptr(), 0, B, T, V, U, loss.data_ptr(), grad.data_ptr(),
                                 _lib.F32, grad.stride(0), grad.stride(1), None, ws.data_ptr(), need, stream)
    assert rc == 0, lib.ctc_amd_last_error()


def timed(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for lib in libs:
    for _ in range(200):
        call(lib)
torch.cuda.synchronize()
res = {n: {"one": [], "p1": [], "p2": []} for n in names}
for r in range(12):
    for n, lib in zip(names, libs):
        call(lib); call(lib)
        res[n]["one"].append(timed(lambda: call(lib)))
        res[n]["p1"].append(timed(lambda: call(lib, False)))
        call(lib, False)
        res[n]["p2"].append(timed(lambda: resume(lib)))
for n in names:
    print(f"{n:>8}: one launch {st.median(res[n]['one']):6.1f} us | phase 1 alone {st.median(res[n]['p1']):6.1f} | phase 2 alone {st.median(res[n]['p2']):6.1f}   (medians of 12 runs of 8 calls, B={B})")
