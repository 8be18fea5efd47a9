"""Diagnostic: interleaved A/B timing of the one-launch loss+gradient call of several builds of the library in ONE process
(same buffers, same clocks, runs of consecutive calls per build, rotating), HIP events around every call; prints median / min per build.
usage: python scripts/ab_time.py tree base swap ...   (names: tree = in-tree library, else scratch/libctc_v_<name>.so)"""
import ctypes, os, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tf_seq2seq_losses_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, U, V = int(os.environ.get("F6_B", "256")), int(os.environ.get("F6_T", "1000")), int(os.environ.get("F6_U", "128")), int(os.environ.get("F6_V", "256"))
kind = 1 if os.environ.get("F6_KINDNAME", "classic") == "simplified" else 0
names = sys.argv[1:] or ["tree"]
libs = []
for n in names:
    path = os.path.join(ROOT, "tf_seq2seq_losses_amd", "libctc_amd.so") if n == "tree" else os.path.join(ROOT, "scratch", f"libctc_v_{n}.so")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)  # (older builds lack the newer entry points)
        if fn is not None:
            fn.restype = restype; fn.argtypes = argtypes
    libs.append(lib)
dev = torch.device("cuda:0")
host, d = bench.make_inputs(B, T, U, V, 2, False, dev)
nbytes = ctypes.c_size_t()
need = 0
for lib in libs:  # (diagnostic builds carry extra workspace regions: the largest request serves all)
    assert lib.ctc_amd_workspace_bytes(_lib.WS_LOSS_GRAD, kind, B, T, V, U, ctypes.byref(nbytes)) == 0
    need = max(need, nbytes.value)
nbytes = ctypes.c_size_t(need)
ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
loss = torch.empty(B, device=dev)
NROT = int(os.environ.get("AB_ROTATE", "1"))  # > 1: that many logits / gradient buffer sets in rotation (nothing of a call's rows is cached)
xs = [d["logits"]] + [d["logits"].clone() for _ in range(NROT - 1)]
grads = [torch.empty(B, T, V, device=dev) for _ in range(NROT)]
stream = torch.cuda.current_stream().cuda_stream
ncall = [0]


def call(lib):
    k = ncall[0] % NROT
    ncall[0] += 1
    rc = lib.ctc_amd_loss_grad(kind, 0, xs[k].data_ptr(), d["labels"].data_ptr(), d["labels"].shape[1], d["label_length"].data_ptr(),
                               d["logit_length"].data_ptr(), 0, B, T, V, U, loss.data_ptr(), grads[k].data_ptr(), None, ws.data_ptr(), nbytes.value, stream)
    assert rc == 0, lib.ctc_amd_last_error()


for lib in libs:
    for _ in range(5):
        call(lib)
torch.cuda.synchronize()
R = int(os.environ.get("AB_ROUNDS", "15"))
RUN = int(os.environ.get("AB_RUN", "12"))   # consecutive calls of one build (its steady state: what the previous call left in the caches is its own)
times = [[] for _ in libs]
for r in range(R):
    for i, lib in enumerate(libs):
        for k in range(RUN):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(lib); e1.record()
            e1.synchronize()
            if k >= 2:
                times[i].append(e0.elapsed_time(e1) * 1e3)
for n, t in zip(names, times):
    t = sorted(t)
    print(f"{n:>10}: median {st.median(t):7.1f} us  p10 {t[len(t) // 10]:7.1f}  min {t[0]:7.1f}  (one loss+gradient call, events; {R} x {RUN} consecutive calls, first 2 of a run dropped)")
