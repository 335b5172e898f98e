#!/usr/bin/env python3
"""Coefficients of the [1, 2] window of csrc/special.h (round 3): psi(z) = (z - z0) g(z) and lnGamma(z) = (z - 1)(z - 2) h(z),
g and h as polynomials in t = z - 1.5 (|t| <= 0.5), fitted at Chebyshev nodes in 60-digit arithmetic and rounded to double.
Prints C arrays and the errors of the fp64 Horner evaluation against mpmath on a dense grid."""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
z0 = mp.findroot(mp.digamma, mp.mpf("1.4616321449683623"))


def fit(f, deg):
    n = deg + 1
    nodes = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]          # on [-1, 1]; t = 0.5 * node
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        t = x / 2
        for j in range(n):
            A[i, j] = t ** j
        b[i] = f(mp.mpf("1.5") + t)
    c = mp.lu_solve(A, b)
    return [float(c[j]) for j in range(n)]


def g(z):
    return mp.digamma(z) / (z - z0) if abs(z - z0) > mp.mpf("1e-25") else mp.polygamma(1, z0)


def h(z):
    d = (z - 1) * (z - 2)
    if abs(z - 1) < mp.mpf("1e-25"):
        return -mp.euler / (z - 2)          # lnGamma'(1) = -gamma
    if abs(z - 2) < mp.mpf("1e-25"):
        return (1 - mp.euler) / (z - 1)     # lnGamma'(2) = 1 - gamma
    return mp.loggamma(z) / d


def horner(c, t):
    p = np.full_like(t, c[-1])
    for a in c[-2::-1]:
        p = p * t + a                       # (no fma on the host model: one more rounding per step than the device)
    return p


z0hi = float(int(z0 * 2 ** 30)) / 2 ** 30                       # 31 significant bits: z - z0hi is exact for z in [1, 2]
z0lo = float(z0 - mp.mpf(z0hi))
for deg in (20, 22, 24):
    cg, ch = fit(g, deg), fit(h, deg)
    zs = np.concatenate([np.linspace(1.0, 2.0, 4001), 1 + np.random.default_rng(1).random(2000)])
    t = zs - 1.5
    psi = ((zs - z0hi) - z0lo) * horner(cg, t)
    lgm = (zs - 1.0) * (zs - 2.0) * horner(ch, t)
    rp = np.array([float(mp.digamma(mp.mpf(float(z)))) for z in zs])
    rl = np.array([float(mp.loggamma(mp.mpf(float(z)))) for z in zs])
    ep = np.max(np.abs(psi - rp))
    el = np.max(np.abs(lgm - rl) / np.maximum(np.abs(rl), 1e-300))
    print(f"// degree {deg}: max |psi error| {ep:.3e} (absolute, |psi| <= 0.58); max lnGamma relative error {el:.3e}")
deg = 22
cg, ch = fit(g, deg), fit(h, deg)
print(f"constexpr double kPsiRootHi = {z0hi!r}, kPsiRootLo = {z0lo!r};")
print("constexpr double kPsiWin[%d] = {%s};" % (deg + 1, ", ".join(repr(c) for c in cg)))
print("constexpr double kLgWin[%d] = {%s};" % (deg + 1, ", ".join(repr(c) for c in ch)))
