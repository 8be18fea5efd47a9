"""ctypes loader of the C restatement (oracle/ctc_oracle.c).  TEST INFRASTRUCTURE ONLY -- same rules as
ctc_oracle.py: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def _load(precision: str):
    if precision not in _libs:
        path = os.path.join(_HERE, f"libctc_oracle_{precision}.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = ctypes.CDLL(path)
        lib.ctc_oracle_loss_grad.restype = ctypes.c_int
        lib.ctc_oracle_loss_grad.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ctc_oracle_num_threads.restype = ctypes.c_int
        _libs[precision] = lib
    return _libs[precision]


def num_threads() -> int:
    return _load("f64").ctc_oracle_num_threads()


def loss_grad(kind, labels, logits, label_length, logit_length, blank_index=0, precision="f64", want_grad=True,
              n_threads=0):
    """Returns (loss[B] float64, grad[B,T,V] float64 or None): gradient of sum(loss) w.r.t. logits."""
    lib = _load(precision)
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    label_length = np.ascontiguousarray(label_length, dtype=np.int32)
    logit_length = np.ascontiguousarray(logit_length, dtype=np.int32)
    B, T, V = logits.shape
    loss = np.empty(B, dtype=np.float64)
    grad = np.empty((B, T, V), dtype=np.float64) if want_grad else None
    rc = lib.ctc_oracle_loss_grad(0 if kind == "classic" else 1, logits.ctypes.data, labels.ctypes.data,
                                  labels.shape[1] if labels.ndim == 2 else 0, label_length.ctypes.data,
                                  logit_length.ctypes.data, int(blank_index), B, T, V, loss.ctypes.data,
                                  grad.ctypes.data if want_grad else None, int(n_threads))
    assert rc == 0
    return loss, grad
