"""One VB-NMF step in 50-digit arithmetic (mpmath): a third, independent statement of reference
src/vbnmf_update.cpp:33-90, used to hold the fp64 oracle (and the HIP engine) to the exact value of the formulas on
small cases.  Test infrastructure only.  Every block cites the reference lines it follows."""
import mpmath as mp
import numpy as np

mp.mp.dps = 50


def _mat(a):
    a = np.asarray(a, dtype=np.float64)
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in a])


def _np(a):
    return np.array([[float(a[i, j]) for j in range(a.cols)] for i in range(a.rows)], dtype=np.float64)


def step(X, wh, hyper, fudge):
    """Returns dict(lw, lh, ew, eh, dw, dh [float64 arrays, correctly rounded from 50 digits], lkh [mpf])."""
    Xm, lw, lh, eh = _mat(X), _mat(wh["lw"]), _mat(wh["lh"]), _mat(wh["eh"])
    n, m, r = Xm.rows, Xm.cols, lw.cols
    aw, bw, ah, bh = (mp.mpf(float(hyper[k])) for k in ("aw", "bw", "ah", "bh"))
    fud = mp.mpf(float(fudge))
    wth = lw * lh                                                         # :33
    xwh = mp.matrix(n, m)
    for i in range(n):
        for j in range(m):
            xwh[i, j] = Xm[i, j] / wth[i, j]                              # :34
    t1, t2 = xwh * lh.T, lw.T * xwh
    alw, bew, ew, dw = mp.matrix(n, r), mp.matrix(n, r), mp.matrix(n, r), mp.matrix(n, r)
    ehsum = [mp.fsum(eh[k, j] for j in range(m)) for k in range(r)]       # :42-43 rowSums of the INCOMING eh
    for i in range(n):
        for k in range(r):
            alw[i, k] = aw + lw[i, k] * t1[i, k]                          # :35, :38-39
            bew[i, k] = aw / bw + ehsum[k]                                # :40-43
            ew[i, k] = alw[i, k] / bew[i, k]                              # :44
            dw[i, k] = alw[i, k] / bew[i, k] / bew[i, k]                  # :46
    alh, beh, ehn, dh = mp.matrix(r, m), mp.matrix(r, m), mp.matrix(r, m), mp.matrix(r, m)
    ewsum = [mp.fsum(ew[i, k] for i in range(n)) for k in range(r)]       # :52-53 colSums of the NEW ew
    for k in range(r):
        for j in range(m):
            alh[k, j] = ah + lh[k, j] * t2[k, j]                          # :36, :48-49
            beh[k, j] = ah / bh + ewsum[k]                                # :50-53
            ehn[k, j] = alh[k, j] / beh[k, j]                             # :54
            dh[k, j] = alh[k, j] / beh[k, j] / beh[k, j]                  # :56
    lwn, lhn = mp.matrix(n, r), mp.matrix(r, m)
    for i in range(n):
        for k in range(r):
            tmp = mp.exp(mp.digamma(alw[i, k])) / bew[i, k]               # :59
            lwn[i, k] = tmp if tmp > fud else fud                         # :60
    for k in range(r):
        for j in range(m):
            tmp = mp.exp(mp.digamma(alh[k, j])) / beh[k, j]               # :63
            lhn[k, j] = tmp if tmp > fud else fud                         # :64
    wth = lwn * lhn                                                       # :67
    A, B = mp.matrix(n, r), mp.matrix(r, m)
    for i in range(n):
        for k in range(r):
            A[i, k] = lwn[i, k] * mp.log(lwn[i, k])                       # :69
    for k in range(r):
        for j in range(m):
            B[k, j] = lhn[k, j] * mp.log(lhn[k, j])                       # :71
    A, B = A * lhn, lwn * B                                               # :70, :72
    eweh = ew * ehn
    U = mp.mpf(0)
    for i in range(n):
        for j in range(m):
            u1 = (A[i, j] + B[i, j]) / wth[i, j] - mp.log(wth[i, j])      # :73-76
            u1 = -eweh[i, j] - Xm[i, j] * u1                              # :77-78
            U += u1 - mp.loggamma(Xm[i, j] + 1)                           # :79-81
    lga = -mp.loggamma(aw) + aw * mp.log(aw / bw)                         # :82
    for i in range(n):
        for k in range(r):
            U += -(aw / bw) * ew[i, k] + lga + alw[i, k] * (1 - mp.log(bew[i, k])) + mp.loggamma(alw[i, k])   # :84-86
    lga = -mp.loggamma(ah) + ah * mp.log(ah / bh)                         # :87
    for k in range(r):
        for j in range(m):
            U += -(ah / bh) * ehn[k, j] + lga + alh[k, j] * (1 - mp.log(beh[k, j])) + mp.loggamma(alh[k, j])  # :88-89
    U /= n * m                                                            # :90
    return {"lw": _np(lwn), "lh": _np(lhn), "ew": _np(ew), "eh": _np(ehn), "dw": _np(dw), "dh": _np(dh), "lkh": U}


CASES = [
    # n, m, r, poisson mean, hyper, fudge, seed
    (7, 9, 3, 0.9, {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 2.220446049250313e-16, 1),
    (12, 8, 2, 0.4, {"aw": 0.05, "bw": 2.0, "ah": 3.0, "bh": 0.3}, 2.220446049250313e-16, 2),   # small shapes: psi far negative
    (6, 15, 5, 2.5, {"aw": 40.0, "bw": 0.7, "ah": 0.5, "bh": 9.0}, 0.0, 3),
    (10, 10, 1, 1.2, {"aw": 1.3, "bw": 0.9, "ah": 0.8, "bh": 1.5}, 1e-3, 4),                    # fudge that clips
]


def make_case(n, m, r, lam, hyper, fudge, seed, noninteger=False):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    if noninteger:
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    wh = {"lw": rng.gamma(hyper["aw"], hyper["bw"] / hyper["aw"], size=(n, r)) + 1e-3,
          "lh": rng.gamma(hyper["ah"], hyper["bh"] / hyper["ah"], size=(r, m)) + 1e-3}
    wh["ew"], wh["eh"] = wh["lw"].copy(), rng.gamma(2.0, 0.5, size=(r, m))
    return np.asfortranarray(X), wh


def ml_step(X, w, h, prior=False, gamma_a=1.0, gamma_b=1.0, eps=2.220446049250313e-16):
    """nmf_updateR + likelihood in 50 digits, reference R/factorize.R:2-27 and :40-49.  Returns (ew, eh, lk)."""
    Xm, W, H = _mat(X), _mat(w), _mat(h)
    n, m, r = Xm.rows, Xm.cols, W.cols
    ga, gb, e = mp.mpf(float(gamma_a)), mp.mpf(float(gamma_b)), mp.mpf(float(eps))

    def ratio(W, H):
        wh = W * H
        q = mp.matrix(n, m)
        for i in range(n):
            for j in range(m):
                q[i, j] = Xm[i, j] / wh[i, j]
        return q

    t = W.T * ratio(W, H)                                                 # :8
    cs = [mp.fsum(W[i, k] for i in range(n)) for k in range(r)]           # :9
    Hn = mp.matrix(r, m)
    for k in range(r):
        for j in range(m):
            up, down = H[k, j] * t[k, j], cs[k]
            if prior:
                up, down = up + ga - 1, down + ga / gb                    # :11-12
            v = up / down                                                 # :14
            Hn[k, j] = e if v < e else v                                  # :15
    t = ratio(W, Hn) * Hn.T                                               # :17 (the NEW h)
    rs = [mp.fsum(Hn[k, j] for j in range(m)) for k in range(r)]          # :18
    Wn = mp.matrix(n, r)
    for i in range(n):
        for k in range(r):
            up, down = W[i, k] * t[i, k], rs[k]
            if prior:
                up, down = up + ga - 1, down + ga / gb                    # :20-21
            v = up / down                                                 # :23
            Wn[i, k] = e if v < e else v                                  # :24
    wh = Wn * Hn                                                          # :42
    lk = mp.mpf(0)
    for i in range(n):
        for j in range(m):
            x = Xm[i, j]
            lk += x * mp.log(wh[i, j]) - wh[i, j]                         # :44
            if x > 0:
                lk += -x * mp.log(x) + x                                  # :45-46
    return _np(Wn), _np(Hn), lk / n / m                                   # :47
