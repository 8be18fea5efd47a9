"""Diagnostic: how much of the second logits read does the Infinity Cache serve?

Times, with HIP events on the launch stream, at the north-star shape:
  * the one-launch loss+gradient call in the bench's steady state (same buffers every call), with a 512 MiB fill before every
    call (cold caches), and rotating over 3 distinct logits/gradient buffer pairs (what a training loop sees: fresh logits);
  * phase 1 alone (loss-only call) and phase 2 alone (ctc_amd_grad_resume) back to back, and with a 512 MiB fill between them.
usage: python scripts/r03_phase_cache.py [lib.so]   -> one JSON object on stdout
"""
import ctypes, json, os, statistics as st, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tf_seq2seq_losses_amd", "libctc_amd.so")
lib = ctypes.CDLL(path)
for name, (restype, argtypes) in _lib.SIGNATURES.items():
    fn = getattr(lib, name, None)
    if fn is not None:
        fn.restype = restype; fn.argtypes = argtypes
B, T, U, V = 256, 1000, 128, 256
kind = 0
dev = torch.device("cuda:0")
sets = []
for seed in (2, 3, 4):
    host, d = bench.make_inputs(B, T, U, V, seed, False, dev)
    d["grad"] = torch.empty(B, T, V, device=dev)
    sets.append(d)
nbytes = ctypes.c_size_t()
assert lib.ctc_amd_workspace_bytes(_lib.WS_LOSS_GRAD, kind, B, T, V, U, ctypes.byref(nbytes)) == 0
ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
loss = torch.empty(B, device=dev)
fillbuf = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def call(d, want_grad=True):
    rc = lib.ctc_amd_loss_grad(kind, 0, d["logits"].data_ptr(), d["labels"].data_ptr(), d["labels"].shape[1], d["label_length"].data_ptr(),
                               d["logit_length"].data_ptr(), 0, B, T, V, U, loss.data_ptr(), d["grad"].data_ptr() if want_grad else None, None,
                               ws.data_ptr(), nbytes.value, stream)
    assert rc == 0, lib.ctc_amd_last_error()


def resume(d):
    x, g = d["logits"], d["grad"]
    rc = lib.ctc_amd_grad_resume(kind, 0, x.data_ptr(), _lib.F32, x.stride(0), x.stride(1), d["labels"].data_ptr(), d["labels"].shape[1],
                                 d["label_length"].data_ptr(), d["logit_length"].data_ptr(), 0, B, T, V, U, loss.data_ptr(), g.data_ptr(),
                                 _lib.F32, g.stride(0), g.stride(1), None, ws.data_ptr(), nbytes.value, stream)
    assert rc == 0, lib.ctc_amd_last_error()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


def med(v):
    v = sorted(v)
    return dict(median_us=round(st.median(v), 1), min_us=round(v[0], 1), p90_us=round(v[int(len(v) * 0.9)], 1), n=len(v))


for _ in range(300):  # ~50 ms of launches: clocks and caches in their steady state
    call(sets[0])
torch.cuda.synchronize()
out = {"lib": os.path.relpath(path, ROOT)}
out["one_launch_same_buffers"] = med([timed(lambda: call(sets[0])) for _ in range(60)])
v = []
for i in range(60):
    v.append(timed(lambda: call(sets[i % 3])))
out["one_launch_rotating_3_buffer_sets"] = med(v[6:])
v = []
for i in range(30):
    fillbuf.fill_(i & 255)
    v.append(timed(lambda: call(sets[0])))
out["one_launch_after_512MiB_fill"] = med(v[3:])
p1, p2 = [], []
for i in range(40):
    p1.append(timed(lambda: call(sets[0], False)))
    p2.append(timed(lambda: resume(sets[0])))
out["phase1_back_to_back"] = med(p1[4:]); out["phase2_back_to_back"] = med(p2[4:])
p1, p2 = [], []
for i in range(30):
    p1.append(timed(lambda: call(sets[0], False)))
    fillbuf.fill_(i & 255)
    p2.append(timed(lambda: resume(sets[0])))
out["phase1_after_phase2"] = med(p1[3:]); out["phase2_after_512MiB_fill"] = med(p2[3:])
p1, p2 = [], []
for i in range(30):
    fillbuf.fill_(i & 255)
    p1.append(timed(lambda: call(sets[0], False)))
    p2.append(timed(lambda: resume(sets[0])))
out["phase1_after_512MiB_fill"] = med(p1[3:]); out["phase2_after_cold_phase1"] = med(p2[3:])
print(json.dumps(out, indent=1))
