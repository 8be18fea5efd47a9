"""Host-side model of the linear-domain chains of csrc/ctc_fused6.hip (test infrastructure).

`Chain` restates Chain<KIND, NL, DIR>::step / renorm / start of the kernel in NumPy float32, lane for lane (64 lanes x NL label
positions, one integer exponent per lane, the upstream neighbour's value rescaled by 2^dk, renormalisation every RN frames with the
adoption rule for lanes without mass), so that what the format loses can be looked at frame by frame against the float64 oracle:
`sweep` runs a chain over the whole utterance and returns its lattice rows as float64 values (mantissa * 2^exponent), in the
reference's index convention (classic: [T+1, L, 2] closed / open, simplified: [T+1, L]), `analyse` compares them with
oracle.ctc_oracle's alpha / beta and prints where posterior mass disappears.

The renormalisation schedule is the one of the kernel's phase 1 (every RN frames inside BLK-frame blocks, blocks counted from each
chain's own end of the utterance); the phase-2 main chains renormalise one frame early and the recompute chains restart from
checkpoints, which changes which frames renormalise, not the rules.

usage: python tests/tools/linear_model.py tests/golden/soak_case_lossonly_u512.npz [NL] [--mask]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

DEAD = -(1 << 24)
GAP, GAP_WIDE = 16, 64
F = np.float32


def cadence(NL):
    """(BLK, RN, LV) of the instantiation with NL label positions per lane at V <= 256 (ctc_fused6.hip: Cad, CTC_F6_ENTRY)"""
    BLK = 12 if NL <= 2 else 6 if NL == 4 else 3
    RN = (6 if NL == 2 else 4) if BLK % 4 == 0 else 3
    return BLK, RN, (RN + NL - 1) // NL


def emissions(x, labels, ll, blank):
    """y[t, i] = exp(x[t, label_i] - rowmax) for i < ll else 0, bl[t] = exp(x[t, blank] - rowmax), l2s[t] = log2 sum_k exp(x - rowmax)"""
    x = np.asarray(x, np.float32)
    mx = x.max(axis=1, keepdims=True)
    e = np.exp2((x * F(1.44269504088896340736) - mx * F(1.44269504088896340736)).astype(np.float32)).astype(np.float32)
    T = x.shape[0]
    y = np.zeros((T, len(labels)), np.float32)
    for i in range(min(ll, len(labels))):
        y[:, i] = e[:, labels[i]] if labels[i] != blank else 0.0
    return y, e[:, blank].copy(), np.log2(e.astype(np.float64).sum(axis=1))


class Chain:
    def __init__(self, kind, NL, DIR, labels, ll, mask_fn=None, gap=GAP, gap_live=16, lift_live=True, full_cascade=False):
        self.kind, self.NL, self.DIR, self.ll = kind, NL, DIR, ll
        self.UP = 64 * NL
        self.BLK, self.RN, self.LV = cadence(NL)
        lab = list(labels[:ll]) + [None] * (self.UP + 2 - ll)
        tok = lambda i: lab[i] if 0 <= i < ll else (-1 - (i < 0))
        idx = np.arange(self.UP)
        self.norep = np.array([(i == 0) or tok(i) != tok(i - 1) for i in idx]).reshape(64, NL)
        self.norep_next = np.array([tok(i + 1) != tok(i) for i in idx]).reshape(64, NL)
        self.c = np.zeros((64, NL), F); self.o = np.zeros((64, NL), F)
        self.cx = F(0); self.k = np.full(64, DEAD, np.int64); self.kx = DEAD; self.dk = np.zeros(64, np.int64)
        self.relevant = (np.arange(64) * NL) <= ll
        self.alive = np.zeros(64, bool); self.age = np.zeros(64, np.int64); self.flag = 0
        self.mask_fn, self.gap, self.lift_live, self.gap_live = mask_fn, gap, lift_live, gap_live  # gap_live=16: the rule until r03
        self.lost = []  # (what, detail) diagnostics
        self.checked = False; self.d9 = []
        self.full_cascade = full_cascade
        self.zprev = np.ones((64, 2 * NL if kind == 0 else NL), bool)

    def start(self):
        if self.DIR == 0:
            self.cx, self.kx = F(1), 0
        else:
            if self.ll == self.UP:
                self.cx, self.kx = F(1), 0
            for i in range(self.UP):
                if i == self.ll:
                    self.c[i // self.NL, i % self.NL] = 1; self.k[i // self.NL] = 0
                if self.kind == 0 and i == self.ll - 1:
                    self.o[i // self.NL, i % self.NL] = 1; self.k[i // self.NL] = 0
        self.renorm()
        self.flag = 0

    def _up(self, v, fill):  # value of the upstream neighbour lane (previous lane for A, next lane for B)
        out = np.empty_like(v)
        if self.DIR == 0:
            out[1:] = v[:-1]; out[0] = fill
        else:
            out[:-1] = v[1:]; out[-1] = fill
        return out

    def step(self, y, bl):
        """y: [64, NL] emissions of the lane's label positions, bl: blank emission"""
        NL, c, o = self.NL, self.c, self.o
        bl = F(bl)
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            if self.kind == 0 and self.DIR == 0:
                m = c + o
                x = np.where(self.norep_next, m, c)
                xin0 = np.ldexp(self._up(x[:, NL - 1], self.cx), np.clip(self.dk, -400, 400).astype(np.int32)).astype(F)
                xin = np.concatenate([xin0[:, None], x[:, :-1]], axis=1)
                self.o = (y * (o + xin)).astype(F)
                self.c = (bl * m).astype(F)
                self.cx = F(self.cx * bl)
            elif self.kind == 0:
                h = (bl * c).astype(F); ee = (y * o).astype(F); pn = h + ee
                x = np.where(self.norep, pn, h)
                self.cx = F(self.cx * bl)
                xinl = np.ldexp(self._up(x[:, 0], self.cx), np.clip(self.dk, -400, 400).astype(np.int32)).astype(F)
                xin = np.concatenate([x[:, 1:], xinl[:, None]], axis=1)
                self.o = (xin + ee).astype(F)
                self.c = pn.astype(F)
            elif self.DIR == 0:
                pin0 = np.ldexp(self._up(c[:, NL - 1], self.cx), np.clip(self.dk, -400, 400).astype(np.int32)).astype(F)
                pin = np.concatenate([pin0[:, None], c[:, :-1]], axis=1)
                self.c = (bl * c + y * pin).astype(F)
                self.cx = F(self.cx * bl)
            else:
                nin = np.ldexp(self._up(c[:, 0], self.cx), np.clip(self.dk, -400, 400).astype(np.int32)).astype(F)
                nx = np.concatenate([c[:, 1:], nin[:, None]], axis=1)
                self.c = (bl * c + y * nx).astype(F)
                self.cx = F(self.cx * bl)

    def renorm(self, t=None):
        if self.mask_fn is not None and t is not None:
            self.mask_fn(self, t)
        m = np.maximum(self.c.max(axis=1), self.o.max(axis=1)) if self.kind == 0 else self.c.max(axis=1)
        live = m > 0
        fe = np.frexp(m)[1].astype(np.int64)
        e_own = np.where(live, fe + self.k, DEAD)
        xlive = self.cx > 0
        ex = int(np.frexp(self.cx)[1]) + self.kx if xlive else DEAD
        kn = e_own.copy()
        if self.lift_live:
            g0 = GAP_WIDE if self.LV == 1 else self.gap
            kn = np.maximum(kn, self._up(kn, ex) - np.where(live, max(self.gap_live, g0), g0))
        else:  # (experiment) only lanes without mass adopt an exponent
            kn = np.where(live, kn, np.maximum(kn, self._up(kn, ex) - (GAP_WIDE if self.LV == 1 else self.gap)))
        if self.full_cascade or (~live & self.relevant).any():
            for _ in range(1, self.LV):
                nb = self._up(kn, ex)
                kn = np.maximum(kn, nb - self.gap) if self.lift_live else np.where(live, kn, np.maximum(kn, nb - self.gap))
        kn = np.maximum(kn, DEAD)
        d = self.k - kn
        self.age = np.where(live & self.alive, self.age + 1, 0)
        self.flag |= (4 if (live & (self.age >= 3) & (d < -96)).any() else 0) | (8 if (live & (fe < -96)).any() else 0) | (16 if (~live & self.alive).any() else 0)
        if self.checked:  # D9 of the kernel: a nonzero mantissa below 2^-100 before or after the shift, or nonzero -> zero
            dneg = np.minimum(d, 0).clip(-400, 0).astype(np.int32)[:, None]
            with np.errstate(under="ignore"):
                vals = np.concatenate([np.ldexp(self.c, dneg), np.ldexp(self.o, dneg)], axis=1).astype(F) if self.kind == 0 else np.ldexp(self.c, dneg).astype(F)
            zeros = (np.concatenate([self.c, self.o], axis=1) == 0) if self.kind == 0 else (self.c == 0)
            tiny = (vals > 0) & (vals < F(2.0 ** -100))
            trans = zeros & ~self.zprev
            if tiny.any() or trans.any():
                self.d9.append((t, np.argwhere(tiny).tolist()[:4], np.argwhere(trans).tolist()[:4], [int(x) for x in d[np.argwhere(tiny)[:4, 0]]] if tiny.any() else []))
            self.zprev = zeros
        dd = np.clip(d, -400, 400).astype(np.int32)[:, None]
        with np.errstate(over="ignore", under="ignore"):
            c2 = np.ldexp(self.c, dd).astype(F); o2 = np.ldexp(self.o, dd).astype(F)
        nz_before = int((self.c > 0).sum() + (self.o > 0).sum()); nz_after = int((c2 > 0).sum() + (o2 > 0).sum())
        if nz_after < nz_before:
            self.lost.append((t, nz_before - nz_after))
        self.c, self.o = c2, o2
        self.k = kn
        if xlive:
            self.cx = F(np.ldexp(self.cx, int(np.clip(self.kx - ex, -400, 400))))
        self.kx = ex
        self.dk = self._up(self.k, self.kx) - self.k
        self.alive = live

    def values(self):
        """(closed[UP], open[UP], cx) as float64 true values"""
        kk = self.k.astype(np.float64)[:, None]
        with np.errstate(over="ignore"):
            sc = np.where(kk <= DEAD / 2, 0.0, np.exp2(np.clip(kk, -5000, 5000)))  # (float64 holds 2^-1074 .. 2^1023; callers pass a log-offset)
        return self.c.astype(np.float64), self.o.astype(np.float64), kk[:, 0], float(self.cx), self.kx


def log2_rows(ch):
    """log2 of the chain's state (base-2 logs survive any exponent): closed[UP], open[UP], cx; -inf where the mantissa is zero"""
    with np.errstate(divide="ignore"):
        lc = np.log2(ch.c.astype(np.float64)) + ch.k[:, None]
        lo = np.log2(ch.o.astype(np.float64)) + ch.k[:, None]
        lx = (np.log2(float(ch.cx)) + ch.kx) if ch.cx > 0 else -np.inf
    lc[ch.c == 0] = -np.inf; lo[ch.o == 0] = -np.inf
    return lc.reshape(-1), lo.reshape(-1), lx


def sweep(kind, NL, DIR, x, labels, ll, tl, blank=0, **kw):
    """Runs one chain over the whole utterance (phase-1 renormalisation schedule).  Returns log2 rows in the reference's index
    convention: classic [T+1, L, 2] (closed, open), simplified [T+1, L]; L = ll + 1; emissions unnormalised (exp(x - rowmax))."""
    BLK, RN, LV = cadence(NL)
    y, bl, l2s = emissions(x[:tl], labels, ll, blank)
    UP = 64 * NL
    ypad = np.zeros((tl, UP), F); ypad[:, :y.shape[1]] = y[:, :UP]
    ch = Chain(kind, NL, DIR, labels, ll, **kw)
    ch.start()
    L = ll + 1
    rows = np.full((tl + 1, L, 2) if kind == 0 else (tl + 1, L), -np.inf)

    def put(t):
        lc, lo, lx = log2_rows(ch)
        if kind == 0:
            if DIR == 0:  # c[i] = closed(l = i + 1), o[i] = open(l = i + 1), cx = closed(0)
                rows[t, 0, 0] = lx
                n = min(L - 1, UP)
                rows[t, 1:1 + n, 0] = lc[:n]; rows[t, 1:1 + n, 1] = lo[:n]
            else:         # c[i] = closed(l = i), o[i] = open(l = i + 1), cx = closed(UP)
                n = min(L, UP)
                rows[t, :n, 0] = lc[:n]
                n1 = min(L - 1, UP)
                rows[t, 1:1 + n1, 1] = lo[:n1]
                if L - 1 == UP: rows[t, UP, 0] = lx
        else:
            if DIR == 0:
                rows[t, 0] = lx; n = min(L - 1, UP); rows[t, 1:1 + n] = lc[:n]
            else:
                n = min(L, UP); rows[t, :n] = lc[:n]
                if L - 1 == UP: rows[t, UP] = lx
    G = (tl + BLK - 1) // BLK
    if DIR == 0:
        put(0)
        for t in range(tl):
            ch.step(ypad[t].reshape(64, NL), bl[t])
            g, d = divmod(t, BLK)
            nv = min(BLK, tl - BLK * g)
            if (d + 1) % RN == 0 or d == nv - 1:
                ch.renorm(t + 1)
            put(t + 1)
    else:
        put(tl)
        for t in range(tl - 1, -1, -1):
            ch.step(ypad[t].reshape(64, NL), bl[t])
            g = t // BLK
            nv = min(BLK, tl - BLK * g)
            d = BLK * g + nv - 1 - t  # position inside the block, counted from the block's upper end
            if (d + 1) % RN == 0 or d == nv - 1:
                ch.renorm(t)
            put(t)
    return rows, l2s, ch


def lse2(a, axis=None):
    a = np.asarray(a, np.float64)
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        return (np.log2(np.exp2(a - m).sum(axis=axis, keepdims=True)) + m).squeeze(axis)


def analyse(path, NL=None, mask=False, verbose=True, **kw):
    from oracle import ctc_oracle as O
    d = np.load(path, allow_pickle=True)
    x, labels, ll, tl = d["x"][0], d["labels"][0], int(d["ll"][0]), int(d["tl"][0])
    kind = int(d["kind"]) if "kind" in d.files else 0
    kind_name = "classic" if kind == 0 else "simplified"
    if NL is None:
        NL = 1 if ll <= 64 else 2 if ll <= 128 else 4 if ll <= 256 else 8
    mask_fn = make_band_mask(kind, ll, tl) if mask else None
    A, l2s, chA = sweep(kind, NL, 0, x, labels, ll, tl, mask_fn=mask_fn, **kw)
    B, _, chB = sweep(kind, NL, 1, x, labels, ll, tl, mask_fn=mask_fn, **kw)
    # float64 oracle on the SAME unnormalised emissions: alpha64 + beta64 per state, log2
    ref = O.ctc_loss(kind_name, labels[None, :max(ll, 1)], x[None, :tl].astype(np.float64), np.array([ll]), np.array([tl]), 0)
    a64 = ref.alpha[0] / np.log(2.0); b64 = ref.beta[0] / np.log(2.0)   # natural logs of normalised emissions -> base 2
    # (normalised emissions: alpha64[t] = alpha_unnorm[t] - sum_{t'<t} l2s, beta64[t] = beta_unnorm[t] - sum_{t'>=t} l2s)
    cs = np.concatenate([[0.0], np.cumsum(l2s)])
    shape = (tl + 1, 1, 1) if kind == 0 else (tl + 1, 1)
    An = A - cs.reshape(shape)
    Bn = B - (cs[-1] - cs).reshape(shape)
    logP64 = -ref.loss[0] / np.log(2.0)
    ax = tuple(range(1, An.ndim))
    mass_lin = np.array([lse2((An[t] + Bn[t]).reshape(-1)) for t in range(tl + 1)])   # log2 sum_s alpha beta per frame, linear chains
    if verbose:
        print(f"{os.path.basename(path)}: {kind_name} NL={NL} ll={ll} tl={tl}; float64 loss {ref.loss[0]:.5f}")
        print(f"  linear chains: per-frame mass / P (should be 1 for every t): min {np.exp2(mass_lin - logP64).min():.6f} max {np.exp2(mass_lin - logP64).max():.6f}")
        tm = ((tl + chA.BLK - 1) // chA.BLK // 2) * chA.BLK
        print(f"  loss from the meeting point tm={tm}: {-mass_lin[tm] * np.log(2.0):.5f} (float64 {ref.loss[0]:.5f}, rel. diff {abs(-mass_lin[tm] * np.log(2.0) - ref.loss[0]) / max(1, abs(ref.loss[0])):.2e})")
        print(f"  soft flags: A {chA.flag} B {chB.flag}; renormalisations that flushed values: A {len(chA.lost)} B {len(chB.lost)}")
        # where does the mass go: states whose float64 posterior is > 1e-6 but whose linear alpha (or beta) is off by > 2^-16
        post = a64 + b64 - logP64
        for name, lin, r64 in (("alpha", An, a64), ("beta", Bn, b64)):
            with np.errstate(invalid="ignore"):
                bad = (post > np.log2(1e-6)) & ~(np.abs(lin - r64) < 2.0 ** -10)
            ts = sorted(set(np.argwhere(bad)[:, 0].tolist()))
            if ts:
                t0 = ts[0] if name == "alpha" else ts[-1]
                st = np.argwhere(bad[t0])
                print(f"  {name}: {int(bad.sum())} states carrying > 1e-6 of the posterior are wrong in the linear chain; first at t={t0}: "
                      + ", ".join(f"state {tuple(int(v) for v in s)} post {np.exp2(post[t0][tuple(s)]):.3g} lin-ref {float(lin[t0][tuple(s)] - r64[t0][tuple(s)]):.3g}" for s in st[:4]))
            else:
                print(f"  {name}: every state carrying > 1e-6 of the posterior agrees with float64 to 2^-10 in log2")
    return np.exp2(mass_lin - logP64), ref.loss[0], chA, chB


def make_band_mask(kind, ll, tl):
    """Exact structural mask: zero what cannot take part in any complete alignment -- alpha at states from which the remaining
    frames cannot emit the remaining labels, beta at states the elapsed frames cannot have reached (products with the other
    direction are zero there anyway; only the lane maximum the renormalisation sees changes)."""
    def fn(ch, t):
        NL = ch.NL
        i = np.arange(ch.UP).reshape(64, NL)
        if ch.DIR == 0:
            # alpha[t, l]: needs ll - l more labels in tl - t frames.  slot i holds l = i + 1 (closed: l labels done; open: in label l)
            rem = tl - t
            dead_c = (ll - (i + 1)) > rem           # closed(l = i+1)
            dead_o = (ll - (i + 1)) > rem           # open(l = i+1): label l is being emitted, ll - l still to come
            ch.c[dead_c] = 0; ch.o[dead_o] = 0
            if ll > rem: ch.cx = F(0)
        else:
            # beta[t, l] needs l labels emitted in t frames: l <= t.  B: c[i] = closed(l = i), o[i] = open(l = i + 1)
            dead_c = i > t
            dead_o = (i + 1) > t
            ch.c[dead_c] = 0; ch.o[dead_o] = 0
            if ch.UP > t: ch.cx = F(0)
    return fn


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    NLa = int(args[1]) if len(args) > 1 else None
    analyse(args[0], NLa, mask="--mask" in sys.argv)
