"""Diagnostic: the default tier (fused6 + fused5 fallback) against the float64 C oracle over a set of shapes; prints the
error, the number of utterances fused6 flagged (read back from the workspace) and the time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops

dev = torch.device("cuda:0")


def flags_of(ws, kind, B, T, V, U):
    o = _lib.flags_offset(kind, B, T, V, U)  # Layout::off_flags of the pipeline's own layout (ctc_amd_debug_flags_offset)
    if os.environ.get("F6_STAMPS"):
        st = ws[o + 4 * B:o + 4 * B + B * 512].view(torch.int64).cpu().numpy().reshape(B, 16, 4)[:, :12]
        med = np.median(st, axis=0)
        names = ["main A", "main B", "rec A", "rec B"] + [f"help A{i}" for i in range(4)] + [f"help B{i}" for i in range(4)]
        print("   per-wave cycles (median over utterances):   phase1 work / wait     phase2 work / wait")
        for w in range(12):
            print(f"     {names[w]:8s} {med[w, 0]:9.0f} {med[w, 1]:9.0f}   {med[w, 2]:9.0f} {med[w, 3]:9.0f}")
    if os.environ.get("F6_DEBUG2"):
        dbg = ws[o + 4 * B:o + 4 * B + B * 2048 * 4].view(torch.int32).cpu().numpy().reshape(B, 2048)[0].reshape(4, 8, 64)
        for d in range(4):
            f = dbg[d].view(np.float32)
            tot = f[0].astype(np.float64).sum() + f[1].astype(np.float64).sum() + f[2].astype(np.float64).sum()
            print(f"   d={d}: total mass {tot / 2**30:.6f}  qb {f[0][:8].tolist()} qt0 {f[1][:8].tolist()} qt1 {f[2][:8].tolist()}")
            print(f"        S.k {dbg[d, 3][:8].tolist()} kR {dbg[d, 4][:8].tolist()} ra0 {f[5][:8].tolist()} ra1 {f[6][:8].tolist()} c0 {f[7][:8].tolist()}")
            big = np.argsort(-(f[0] + f[1] + f[2]))[:6]
            print(f"        heaviest lanes {big.tolist()} mass {((f[0] + f[1] + f[2])[big] / 2**30).tolist()}")
        hd = ws[o + 4 * B:o + 4 * B + B * 2048 * 4].view(torch.int32).cpu().numpy().reshape(B, 2048)[0][1024:1024 + 192].reshape(3, 64).view(np.float32)
        md = dbg[1].view(np.float32)
        print("   helper-1 read of d=1 equals what main wrote:", np.array_equal(hd[0], md[0]), np.array_equal(hd[1], md[1]), np.array_equal(hd[2], md[2]))
        bad = np.nonzero((hd[0] != md[0]) | (hd[1] != md[1]) | (hd[2] != md[2]))[0]
        print("   differing lanes", bad.tolist()[:20], "helper", hd[:, bad[:6]].tolist(), "main", md[:3, bad[:6]].tolist())
    if os.environ.get("F6_DEBUG"):
        fl = ws[o:o + 4 * B].view(torch.int32).cpu().numpy()
        dbg = ws[o + 4 * B:o + 4 * B + B * 2048 * 4].view(torch.int32).cpu().numpy().reshape(B, 2048)[:, :512].reshape(B, 2, 4, 64)
        shown = 0
        for b in range(B):
            for d in range(2):
                lanes = np.nonzero(dbg[b, d, 0])[0]
                if len(lanes) and shown < 12:
                    shown += 1
                    print(f"   b={b} dir={d} flag={hex(fl[b])} D3 lanes {lanes.tolist()} renorm# {dbg[b, d, 0, lanes].tolist()} d {dbg[b, d, 1, lanes].tolist()} "
                          f"fe {dbg[b, d, 2, lanes].tolist()} k_end(lanes 0..63) {dbg[b, d, 3].tolist()}")
    return ws[o:o + 4 * B].view(torch.int32).cpu().numpy()


def run(kind, B, T, U, V, seed=0, ragged=False, scale=1.0, ncheck=4, reps=20, tweak=None):
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    labels = rng.integers(1, V, (B, U), dtype=np.int32)
    if ragged:
        tl = rng.integers(T // 2, T + 1, B).astype(np.int32); ll = rng.integers(U // 2, U + 1, B).astype(np.int32)
    else:
        tl = np.full(B, T, np.int32); ll = np.full(B, U, np.int32)
    if tweak:
        tweak(logits, labels, ll, tl)
    k = ops.KINDS[kind]
    p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(logits).to(dev), torch.from_numpy(ll).to(dev),
                     torch.from_numpy(tl).to(dev), 0, U=U)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, k, B, T, V, U), dtype=torch.uint8, device=dev)
    name = _lib.pipeline_name(k, 0, B, T, V, U, True)
    loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
    torch.cuda.synchronize()
    fl = flags_of(ws, k, B, T, V, U) if name == "fused6" else np.zeros(B, np.int32)
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = min(B, ncheck)
    rl, rg = C.loss_grad(kind, labels[:n], logits[:n], ll[:n], tl[:n], 0)
    ln, gn = loss.cpu().numpy(), grad.cpu().numpy()
    fin = np.isfinite(rl)
    okfin = np.array_equal(np.isfinite(ln[:n]), fin)
    lerr = (np.abs(ln[:n][fin] - rl[fin]) / np.maximum(1, np.abs(rl[fin]))).max() if fin.any() else 0.0
    gerr = np.abs(gn[:n] - rg).max()
    if gerr > 1e-3:
        e = np.abs(gn[:n] - rg)
        bad = np.argwhere(e.max(axis=2) > 1e-3)
        print("   bad frames of b=0:", [int(t) for bb, t in bad if bb == 0], "tl", tl[:n].tolist(), "ll", ll[:n].tolist())
        b0, t0 = bad[0]
        ks = np.argsort(-e[b0, t0])[:6]
        print("   worst tokens at first bad frame:", [(int(k), float(gn[b0, t0, k]), float(rg[b0, t0, k])) for k in ks], "labels", labels[b0][:8].tolist())
    if np.isnan(gn).any():
        bad = np.argwhere(np.isnan(gn).any(axis=2))
        print("   NaN frames (b, t):", bad[:24].tolist(), "tl", tl[:4].tolist())
    print(f"{name:7s} {kind:10s} B={B:4d} T={T:5d} U={U:4d} V={V:5d} {'ragged' if ragged else 'full  '} scale={scale:g}: "
          f"finite-match={okfin} loss rel {lerr:.2e} grad {gerr:.2e} nan={int(np.isnan(gn).sum())} flagged={int((fl != 0).sum())}/{B} "
          f"(bits {hex(int(np.bitwise_or.reduce(fl))) if B else 0}) {dt * 1e3:.3f} ms", flush=True)
    return lerr, gerr


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        for kind in ("classic", "simplified"):
            run(kind, 4, 40, 6, 256, ragged=True)
            run(kind, 6, 70, 20, 256, ragged=True)
            run(kind, 6, 150, 100, 256, ragged=True)
            run(kind, 6, 97, 40, 512, ragged=True)
            run(kind, 6, 40, 12, 1024, ragged=True)
            run(kind, 6, 260, 200, 256, ragged=True)
            run(kind, 5, 33, 7, 29, ragged=True)
            run(kind, 3, 300, 64, 64, ragged=False)
    if which == "simp":
        run("simplified", 2, 150, 100, 256, ragged=False)
        run("simplified", 2, 48, 100, 256, ragged=False)
        run("simplified", 2, 150, 65, 256, ragged=False)
    if which == "seeds":
        for sd in range(int(os.environ.get("F6_NSEEDS", "8"))):
            run("classic", 256, 1000, 128, 256, seed=sd, ncheck=2, reps=10)
    if which == "stride":
        # same work (logit_length = 1000) with different allocation strides between utterances: T_alloc rows of 1 KB each
        for Talloc in (1000, 1001, 1003, 1008, 1024):
            def tw(logits, labels, ll, tl):
                tl[:] = 1000
            run("classic", 256, Talloc, 128, 256, seed=2, ncheck=1, reps=50, tweak=tw)
    if which == "b128":
        run("classic", 128, 1000, 128, 256, seed=2, ncheck=2, reps=50)
        run("classic", 192, 1000, 128, 256, seed=2, ncheck=2, reps=50)
        run("classic", 256, 1000, 128, 256, seed=2, ncheck=2, reps=50)
    if which == "sweepcase":
        T, U, V, B = (int(a) for a in sys.argv[2:6])
        kind = sys.argv[6] if len(sys.argv) > 6 else "classic"
        rng = np.random.default_rng(T * 1000 + U * 10 + V)
        z = rng.standard_normal((B, T, V))
        sc = rng.choice([0.5, 1.0, 4.0])
        logits = (z * sc).astype(np.float32)
        labels = rng.integers(1, V, (B, max(U, 1))).astype(np.int32)
        if U >= 4:
            labels[0, : U // 2] = labels[0, 0]
        ll = rng.integers(0, U + 1, B).astype(np.int32)
        tl = rng.integers(0, T + 1, B).astype(np.int32)
        ll[0], tl[0] = U, T
        print("scale", sc, "ll", ll.tolist(), "tl", tl.tolist())
        k = ops.KINDS[kind]
        p = ops.Prepared(torch.from_numpy(labels).to(dev), torch.from_numpy(logits).to(dev), torch.from_numpy(ll).to(dev),
                         torch.from_numpy(tl).to(dev), 0, U=U)
        rl, rg = C.loss_grad(kind, labels, logits, ll, tl, 0)
        loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True)
        torch.cuda.synchronize()
        e0 = np.abs(grad.cpu().numpy() - rg).max(axis=(1, 2))
        loss_only, _ = ops.loss_grad(k, _lib.WRT_LOGITS, p, False)
        torch.cuda.synchronize()
        e1 = np.abs(grad.cpu().numpy() - rg).max(axis=(1, 2))
        print("default workspace path: per-utterance grad err right after the call", e0.tolist(), "after a loss-only call", e1.tolist(),
              "loss", loss.cpu().numpy().tolist(), "loss-only", loss_only.cpu().numpy().tolist())
        for pipe in ("", "fused5", "v1"):
            _lib.debug_override("pipeline", pipe)
            nws = _lib.workspace_bytes(_lib.WS_LOSS_GRAD, k, B, T, V, U)
            for fill in ("zeros", "ones", "random"):
                ws = (torch.zeros(nws, dtype=torch.uint8, device=dev) if fill == "zeros" else
                      torch.full((nws,), 255, dtype=torch.uint8, device=dev) if fill == "ones" else
                      torch.randint(0, 256, (nws,), dtype=torch.uint8, device=dev))
                loss, grad = ops.loss_grad(k, _lib.WRT_LOGITS, p, True, workspace=ws)
                torch.cuda.synchronize()
                print(f"   workspace filled with {fill}: max grad err {np.abs(grad.cpu().numpy() - rg).max():.3e}")
            gn = grad.cpu().numpy()
            err = np.abs(gn - rg).max(axis=(1, 2))
            fl = flags_of(ws, k, B, T, V, U) if pipe == "" else None
            print(f"pipeline {pipe or 'default'}: per-utterance grad err {err.tolist()} loss {loss.cpu().numpy().tolist()} ref {rl.tolist()} flags {None if fl is None else [hex(int(f)) for f in fl]}")
            if pipe == "":
                b0 = int(np.argmax(err))
                e = np.abs(gn[b0] - rg[b0])
                bad = np.nonzero(e.max(axis=1) > 1e-4)[0]
                print("   worst utterance", b0, "frames with err > 1e-4:", bad.tolist(), "max |row sum of grad|:", np.abs(gn[b0].sum(axis=1)).max())
                for t in bad[:4]:
                    ks = np.argsort(-e[t])[:4]
                    print(f"     t={t}: row sum {gn[b0, t].sum():.3e}; worst tokens (k, got, want):", [(int(kk), float(gn[b0, t, kk]), float(rg[b0, t, kk])) for kk in ks])
        _lib.debug_override("pipeline", "")
    if which == "scal":
        for (U, V) in ((128, 256), (64, 256), (128, 128), (64, 128), (32, 64), (128, 252), (20, 256)):
            run("classic", 256, 1000, U, V, seed=2, ncheck=1, reps=50)
    if which == "nsc":
        run("classic", 256, 1000, 128, 256, seed=int(os.environ.get("F6_SEED", "2")), ncheck=2, reps=50)
    if which == "ns":
        run("classic", 256, 1000, 128, 256, ncheck=8)
        run("simplified", 256, 1000, 128, 256, ncheck=8)
    if which in ("all", "big"):
        for kind in ("classic", "simplified"):
            run(kind, 256, 1000, 128, 256, ncheck=8)
            run(kind, 256, 1000, 128, 256, ragged=True, seed=1, ncheck=8)
            run(kind, 256, 1000, 128, 256, scale=3.0, ncheck=4)
            run(kind, 256, 1000, 128, 256, scale=10.0, ncheck=4)
        run("classic", 64, 5000, 128, 256, ncheck=2, reps=5)
        run("classic", 256, 500, 256, 256, ncheck=4)
        run("classic", 256, 500, 60, 512, ncheck=4)
        run("classic", 256, 300, 100, 1024, ncheck=4)
        run("classic", 512, 1000, 128, 256, ragged=True, ncheck=4)
    if which == "long":
        for kind in ("classic", "simplified"):
            for U in (257, 300, 400, 512):
                run(kind, 256, 1000, U, 256, seed=1, ncheck=2, reps=20)
        run("classic", 64, 2000, 512, 256, seed=1, ncheck=1, reps=5)
        run("classic", 256, 1000, 512, 512, seed=1, ncheck=1, reps=10)
    if which == "longT":
        for kind in ("classic", "simplified"):
            run(kind, 64, 5000, 128, 256, seed=0, ncheck=1, reps=3)
            run(kind, 256, 4000, 128, 256, seed=1, ragged=True, ncheck=1, reps=3)
            run(kind, 256, 3000, 100, 256, seed=2, ragged=True, ncheck=1, reps=3)
