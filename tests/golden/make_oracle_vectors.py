"""Writes tests/golden/oracle_vectors.npz: seeded random inputs with fp64 oracle outputs.

SELF-GENERATED, NOT REFERENCE-GENERATED: the TensorFlow reference cannot run here, so these vectors come from
oracle/ctc_oracle.py after it passed the reference's known-answer table, brute force, torch and finite-difference
checks (tests/test_oracle_*.py).  They freeze the oracle's behaviour so that an accidental change of the oracle
cannot silently move the target the HIP path is compared with.
Run:  python tests/golden/make_oracle_vectors.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ctc_oracle as O  # noqa: E402

CASES = [  # (name, B, T, V, seed, max_label_length or None)
    ("ref_sizes_a", 8, 20, 8, 0, None),
    ("ref_sizes_b", 8, 64, 10, 1, None),
    ("tiny_hessian", 2, 4, 2, 3, None),
    ("wide_label", 3, 150, 33, 70, 70),
]

if __name__ == "__main__":
    out = {}
    for name, B, T, V, seed, U in CASES:
        inp = O.generate_ctc_loss_inputs(B, T, seed, V, max_label_length=U)
        for k in ("labels", "logits", "label_length", "logit_length"):
            out[f"{name}/{k}"] = inp[k]
        for kind in ("classic", "simplified"):
            d = O.ctc_loss(kind, inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
            out[f"{name}/{kind}/loss"] = d.loss
            out[f"{name}/{kind}/grad_logits"] = O.logits_gradient(d, inp["logits"])
            if T <= 20:
                out[f"{name}/{kind}/hessian_logits"] = O.logits_hessian(d, inp["logits"]).astype(np.float32)  # 1e-4 tolerance: float32 storage
                out[f"{name}/{kind}/hessian_logprobs"] = d.hessian.astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
