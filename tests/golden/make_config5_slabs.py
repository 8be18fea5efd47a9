"""Writes tests/golden/config5_hessian_slabs.npz: entries of the float64 oracle Hessian at the FULL size of BASELINE.json
configs[4] (classic / simplified, T=200 U=32 V=64), for utterances 0 and 13 of the config-5 tensor.

SELF-GENERATED, NOT REFERENCE-GENERATED (TensorFlow is not available here): the vectors come from oracle/ctc_oracle.py --
the line-by-line NumPy restatement of base_loss.py:186-260 with the gamma scans of classic_ctc_loss.py:167-308 and
simplified_ctc_loss.py:85-191 -- after it passed the reference's known answers, brute force and finite differences
(tests/test_oracle_*.py).  One utterance at T=200 needs gamma [T+1,L,2,T+1,L,2] = 1.4 GB and a [T,V,T,V] Hessian of 1.3 GB
in float64; ~10 GB peak, a few minutes on the build container.  Stored: for each lattice and utterance 20 slabs
H[t1,k1,:,:] ([T,V] each) in logits space and in log-probability space, float32 (the comparison tolerance is 1e-4), with the
(t1,k1) list.  Inputs are regenerated from the seed in the test (tests/test_gpu_configs.py::_inputs), not stored.

Run:  python tests/golden/make_config5_slabs.py        (writes ~1.6 MB)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ctc_oracle as O  # noqa: E402

B, T, U, V = 32, 200, 32, 64
UTTERANCES = (0, 13)
N_SLABS = 20


def inputs(seed=0):
    """tests/test_gpu_configs.py::_inputs(32, 200, 32, 64, 0)"""
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((B, T, V), dtype=np.float32)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    return logits, labels, np.full(B, U, dtype=np.int32), np.full(B, T, dtype=np.int32)


def slab_index(labels_b, rng):
    """20 (t1, k1) pairs: the first and last frame, the blank, tokens present in the label and absent ones"""
    present = sorted(set(int(k) for k in labels_b))
    absent = [k for k in range(1, V) if k not in present]
    picks = [(0, 0), (T - 1, 0), (0, present[0]), (T - 1, present[-1]), (T // 2, 0), (T // 2, absent[0])]
    while len(picks) < N_SLABS:
        t1 = int(rng.integers(0, T))
        k1 = int(rng.choice(present if len(picks) % 3 else [0] + absent[:3]))
        if (t1, k1) not in picks:
            picks.append((t1, k1))
    return np.array(picks, dtype=np.int32)


if __name__ == "__main__":
    logits, labels, ll, tl = inputs()
    out = {}
    rng = np.random.default_rng(5)
    for b in UTTERANCES:
        idx = slab_index(labels[b], rng)
        out[f"u{b}/index"] = idx
        for kind in ("classic", "simplified"):
            t0 = time.time()
            d = O.ctc_loss(kind, labels[b:b + 1], logits[b:b + 1], ll[b:b + 1], tl[b:b + 1], 0)
            h_lp = d.hessian[0]                                   # [T,V,T,V] w.r.t. log-probabilities (base_loss.py:186-260)
            out[f"u{b}/{kind}/logprobs"] = np.stack([h_lp[t1, k1] for t1, k1 in idx]).astype(np.float32)
            h_x = O.logits_hessian(d, logits[b:b + 1])[0]         # w.r.t. logits (README.md:58-71)
            out[f"u{b}/{kind}/logits"] = np.stack([h_x[t1, k1] for t1, k1 in idx]).astype(np.float32)
            out[f"u{b}/{kind}/loss"] = d.loss.astype(np.float64)
            print(f"utterance {b} {kind}: {time.time() - t0:.0f} s, loss {d.loss[0]:.4f}, max|H_logits| {np.abs(h_x).max():.3f}", flush=True)
            del d, h_lp, h_x
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config5_hessian_slabs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
