"""The reference's README benchmark table on MI355X (tests/benchmark.py:36-237, README.md:16-24).

Same experiment: batch 256, 32 tokens, 255 frames, ragged lengths in the distribution of tests/common.py:77-94
(logit_length ~ U{T/2..T-1}, label_length ~ U{T/4..T/2-1}, label tensor as wide as T), 3 warm-up + 10 timed steps,
forward and forward+gradient, for the framework's built-in CTC (there: tf.nn.ctc_loss; here:
torch.nn.functional.ctc_loss on the same GPU), classic_ctc_loss and simple_ctc_loss.  Two rows the reference cannot
time at this size are added: the second-order product (README.md:58-71 contracted with a vector) through ctc_amd_hvp.

    python benchmarks/reference_table.py [--steps 10 --warmup 3 --json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_seq2seq_losses_amd as ctc  # noqa: E402


def make_inputs(B, T, V, seed, dev):
    rng = np.random.default_rng(seed)
    logits = torch.from_numpy(rng.standard_normal((B, T, V), dtype=np.float32)).to(dev)
    logit_length = torch.from_numpy(rng.integers(T // 2, T, B, dtype=np.int32)).to(dev)
    label_length = torch.from_numpy(rng.integers(T // 4, T // 2, B, dtype=np.int32)).to(dev)
    labels = torch.from_numpy(rng.integers(1, V, (B, T), dtype=np.int32)).to(dev)
    return labels, logits, label_length, logit_length


def torch_ctc(labels, logits, label_length, logit_length, blank_index=0):
    lp = torch.log_softmax(logits, dim=2).transpose(0, 1)
    return torch.nn.functional.ctc_loss(lp, labels.long(), logit_length.long(), label_length.long(), blank=blank_index,
                                        reduction="none", zero_infinity=False)


def timeit(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.mean(ts)), float(np.std(ts))


def tsweep(a, dev):
    """Length sweep: the reference's ragged distribution at B = 256 for growing T, labels growing
    with T (README.md:38-43 "gradient complexity O(l^2)": frames x label positions) and labels fixed at U = 128 (then the
    lattice is linear in T).  Forward + gradient through the public functions; ms per call and ns per lattice cell."""
    from tf_seq2seq_losses_amd import _lib, ops
    rows = []
    for V in (32, 256):
        for mode in ("labels ~ T", "labels <= 128"):
            for T in (64, 128, 255, 500, 1000, 2000, 4000):
                rng = np.random.default_rng(T + V)
                tl_np = rng.integers(T // 2, T, a.B, dtype=np.int32)
                hi = min(T // 2, 1000) if mode == "labels ~ T" else min(T // 2, 128)
                ll_np = rng.integers(max(hi // 2, 1), max(hi, 2), a.B, dtype=np.int32)
                U = int(ll_np.max())
                logits = torch.from_numpy(rng.standard_normal((a.B, T, V), dtype=np.float32)).to(dev)
                labels = torch.from_numpy(rng.integers(1, V, (a.B, U), dtype=np.int32)).to(dev)
                ll, tl = torch.from_numpy(ll_np).to(dev), torch.from_numpy(tl_np).to(dev)
                for name, fn, kind in (("classic_ctc_loss", ctc.classic_ctc_loss, 0), ("simple_ctc_loss", ctc.simple_ctc_loss, 1)):
                    def gradient():
                        x = logits.detach().requires_grad_(True)
                        loss = fn(labels, x, ll, tl, 0)
                        return torch.autograd.grad(torch.where(torch.isfinite(loss), loss, 0.0).sum(), x)[0]
                    ms, sd = timeit(gradient, a.steps, a.warmup)
                    cells = float((tl_np.astype(np.int64) * (2 * ll_np.astype(np.int64) + 1)).sum())
                    rows.append({"V": V, "labels": mode, "T": T, "U": U, "name": name, "ms": ms, "std": sd,
                                 "pipeline": _lib.pipeline_name(kind, 0, a.B, T, V, U, True),
                                 "ns_per_cell": ms * 1e6 / cells, "ns_per_frame": ms * 1e6 / float(tl_np.sum())})
                del logits
    print(f"B={a.B}, ragged lengths (logit_length ~ U[T/2, T), label_length ~ U[hi/2, hi)), {a.steps} steps after {a.warmup} warm-up, "
          f"forward + gradient, wall clock per call incl. Python, {torch.cuda.get_device_name(0)}")
    print("| V | labels | T | max label | function | pipeline | ms | ns per frame | ns per lattice cell |")
    print("|--:|:--|--:|--:|:--|:--|--:|--:|--:|")
    for r in rows:
        print(f"| {r['V']} | {r['labels']} | {r['T']} | {r['U']} | `{r['name']}` | {r['pipeline']} | {r['ms']:.3f} | {r['ns_per_frame']:.2f} | {r['ns_per_cell']:.4f} |")
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"config": vars(a), "device": torch.cuda.get_device_name(0), "rows": rows}, fh, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tsweep", action="store_true", help="length sweep instead of the README table")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--T", type=int, default=255)
    ap.add_argument("--V", type=int, default=32)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.tsweep:
        return tsweep(a, dev)
    labels, logits, ll, tl = make_inputs(a.B, a.T, a.V, 0, dev)
    v = torch.randn_like(logits)
    impls = {"torch.nn.functional.ctc_loss": torch_ctc, "classic_ctc_loss": ctc.classic_ctc_loss,
             "simple_ctc_loss": ctc.simple_ctc_loss}
    rows = {}
    for name, fn in impls.items():
        def forward():
            with torch.no_grad():
                return fn(labels, logits, ll, tl, 0)

        def gradient():
            x = logits.detach().requires_grad_(True)
            loss = fn(labels, x, ll, tl, 0)
            return torch.autograd.grad(torch.where(torch.isfinite(loss), loss, 0.0).sum(), x)[0]

        def second():
            x = logits.detach().requires_grad_(True)
            loss = fn(labels, x, ll, tl, 0)
            (g,) = torch.autograd.grad(torch.where(torch.isfinite(loss), loss, 0.0).sum(), x, create_graph=True)
            return torch.autograd.grad((g * v).sum(), x)[0]

        rows[name] = {"forward_ms": timeit(forward, a.steps, a.warmup), "gradient_ms": timeit(gradient, a.steps, a.warmup)}
        if name != "torch.nn.functional.ctc_loss":  # torch's CTC has no second derivative
            rows[name]["hvp_ms"] = timeit(second, a.steps, a.warmup)
    print(f"B={a.B} T={a.T} V={a.V}, {a.steps} steps after {a.warmup} warm-up, wall clock per call incl. Python, "
          f"{torch.cuda.get_device_name(0)}")
    print("| Name | Forward (ms) | Forward + gradient (ms) | + Hessian-vector product (ms) |")
    print("|:--|:-:|:-:|:-:|")
    for name, r in rows.items():
        f = lambda k: ("%.3g ± %.1g" % r[k]) if k in r else "n/a"
        print(f"| `{name}` | {f('forward_ms')} | {f('gradient_ms')} | {f('hvp_ms')} |")
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"config": vars(a), "device": torch.cuda.get_device_name(0), "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
