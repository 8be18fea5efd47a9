"""Where does the gradient error of a log-domain CTC sweep come from?  Classic lattice, one utterance, alpha / beta in base-2 logs
with numpy; emission precision and chain precision varied independently; posteriors normalised by the frame mass as the kernels do.
usage: python tests/tools/logdomain_error_model.py [T] [sigma]   (CPU only; r04: float32 chain 2e-5 ... 8e-5, float64 chain 9e-7)"""
import numpy as np, sys
rng = np.random.default_rng(0)
T, U, V, sigma = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 128, 256, float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
x = (rng.standard_normal((T, V)) * sigma).astype(np.float32)
lab = rng.integers(1, V, U)
S = 2 * U + 1
ext = np.zeros(S, np.int64); ext[1::2] = lab
skip = np.zeros(S, bool); skip[3::2] = lab[1:] != lab[:-1]

def emissions(dt_e):
    xe = x.astype(np.float64)
    lp = (xe - xe.max(1, keepdims=True)); lp = (lp - np.log(np.exp(lp).sum(1, keepdims=True))) / np.log(2)
    if dt_e == np.float32:  # as the kernel: float32 arithmetic
        xf = x; m = xf.max(1, keepdims=True); l2s = np.log2(np.exp2((xf - m) * np.float32(1.4426950408889634)).sum(1, keepdims=True, dtype=np.float32)).astype(np.float32)
        lp = ((xf - m) * np.float32(1.4426950408889634) - l2s).astype(np.float32)
    return lp[:, ext].astype(dt_e)

def lse(a, b, dt):
    m = np.maximum(a, b); d = -np.abs(a - b)
    return (m + np.log2(dt(1) + np.exp2(d).astype(dt)).astype(dt)).astype(dt)

def sweep(E, dt, renorm_every):
    NEG = dt(-1e30)
    A = np.empty((T, S), np.float64); offA = 0.0
    a = np.full(S, NEG, dt); a[0] = E[0, 0]; a[1] = E[0, 1]
    A[0] = a
    for t in range(1, T):
        p1 = np.concatenate(([NEG], a[:-1])); p2 = np.concatenate(([NEG, NEG], a[:-2])); p2 = np.where(skip, p2, NEG)
        a = (lse(lse(a, p1, dt), p2, dt) + E[t]).astype(dt)
        if t % renorm_every == 0:
            m = a.max(); a = (a - m).astype(dt); offA += float(m)
        A[t] = a.astype(np.float64) + offA
    Bm = np.empty((T, S), np.float64); offB = 0.0
    b = np.full(S, NEG, dt); b[S - 1] = 0; b[S - 2] = 0
    Bm[T - 1] = b
    for t in range(T - 2, -1, -1):
        be = (b + E[t + 1]).astype(dt)
        n1 = np.concatenate((be[1:], [NEG])); n2 = np.concatenate((be[2:], [NEG, NEG])); n2 = np.where(np.concatenate((skip[2:], [False, False])), n2, NEG)
        b = lse(lse(be, n1, dt), n2, dt)
        if t % renorm_every == 0:
            m = b.max(); b = (b - m).astype(dt); offB += float(m)
        Bm[t] = b.astype(np.float64) + offB
    return A, Bm

def posterior(A, Bm):
    q = A + Bm
    q = np.exp2(q - q.max(1, keepdims=True)); q /= q.sum(1, keepdims=True)   # normalised by the frame's own mass
    P = np.zeros((T, V))
    for s in range(S): P[:, ext[s]] += q[:, s]
    return P

ref = posterior(*sweep(emissions(np.float64), np.float64, 1))
for name, de, dc, rn in (("emissions f32, chain f32, renorm 12", np.float32, np.float32, 12), ("emissions f32, chain f32, renorm 1", np.float32, np.float32, 1),
                         ("emissions f64, chain f32, renorm 1", np.float64, np.float32, 1), ("emissions f32, chain f64", np.float32, np.float64, 1)):
    E = emissions(de)
    P = posterior(*sweep(E.astype(dc) if dc == np.float32 else E.astype(np.float64), dc, rn))
    print(f"T {T} sigma {sigma}: {name:40s} max |d posterior| {np.abs(P - ref).max():.2e}", flush=True)
