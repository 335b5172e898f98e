// special.h -- fp64 special functions for gfx950 device code: ln, digamma, log-gamma.
//
// The reference evaluates psi and ln Gamma through GSL (gsl_sf_psi, gsl_sf_lngamma; reference
// src/vbnmf_update.cpp:59,63,82,85,87,89).  HIP has no digamma, ocml's log costs ~95
// double-double instructions, and the engine only needs positive arguments
// (alw = aw + sw >= aw > 0), so all three are written out here, branch-free.
//
// The functions are __host__ __device__ so the same source can be checked on the CPU build
// against mpmath (tests/test_special_cpu.py through vbnmf_test_special); the device build is
// checked on the GPU (tests/test_gpu_special.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace vbnmf {

// ---- building blocks that have a gfx950 instruction on the device and a libm form on the host
__host__ __device__ __forceinline__ double sp_rcp_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    double r = 1.0 / x;                        // deliberately coarsened to ~24 bits: the Newton steps must cope
    unsigned long long u;
    __builtin_memcpy(&u, &r, 8);
    u &= ~((1ULL << 29) - 1);
    __builtin_memcpy(&r, &u, 8);
    return r;
#endif
}
__host__ __device__ __forceinline__ double sp_frexp(double x, int *k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *k = __builtin_amdgcn_frexp_exp(x);
    return __builtin_amdgcn_frexp_mant(x);
#else
    return std::frexp(x, k);
#endif
}

// 1/w to full precision: seed (v_rcp_f64 is good to 2^-24, measured) plus two Newton steps.
__host__ __device__ __forceinline__ double sp_rcp(double w)
{
    double rc = sp_rcp_seed(w);
    rc = fma(fma(-w, rc, 1.0), rc, rc);
    rc = fma(fma(-w, rc, 1.0), rc, rc);
    return rc;
}

// x / w for finite w of ordinary magnitude (no range scaling: the engine's divisors are far from
// the exponent limits): seed, ONE Newton step (2^-48), quotient, one residual correction of the
// quotient (which squares the error again): < 1 ulp.
__host__ __device__ __forceinline__ double dev_div(double x, double w)
{
    double rc = sp_rcp_seed(w);
    rc = fma(fma(-w, rc, 1.0), rc, rc);
    const double q = x * rc;
    return fma(fma(-w, q, x), rc, q);
}

// x / w to 2^-48 relative (one-sided: the result is (x/w)(1 - e^2), e = the seed's relative error <= 2^-24.4):
// seed, quotient, ONE residual correction -- 3 fp64 operations after the seed instead of 5.  For the sweep's
// q = x / wth, whose consumers are sums held to 1e-12.
__host__ __device__ __forceinline__ double dev_div_fast(double x, double w)
{
    const double rc = sp_rcp_seed(w);
    const double q = x * rc;
    return fma(fma(-w, q, x), rc, q);
}

// ---- table-driven ln for the sweep's inner loop ------------------------------------------
// x = m 2^k, m in [0.5, 1) cut into 128 intervals; per interval c = 1/midpoint (rounded) and
// -ln(c); r = m c - 1 is exact in one fma and |r| <= 2^-8, so
//     ln x = k ln2 - ln c + (r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6)        (|r|^7/7 < 2e-18)
// ~15 instructions and a 16-byte table read instead of ~35 instructions with a division.
constexpr int kLogTabSize = 128;
struct LogTabEntry { double c, neg_log_c; };
inline void fill_log_table(LogTabEntry *t)
{
    for (int i = 0; i < kLogTabSize; i++) {
        const long double mid = 0.5L + (i + 0.5L) / 256.0L;
        const double c = (double)(1.0L / mid);
        t[i].c = c;
        t[i].neg_log_c = (double)(-logl((long double)c));
    }
}
__host__ __device__ __forceinline__ double dev_log_tab(double x, const LogTabEntry *__restrict__ tab)
{
    int k;
    const double m = sp_frexp(x, &k);                     // [0.5, 1)
    unsigned long long u;
    __builtin_memcpy(&u, &m, 8);
    const unsigned idx = (unsigned)(u >> 45) & 127u;      // top 7 mantissa bits
    const double c = tab[idx].c, lc = tab[idx].neg_log_c;
    const double r = fma(m, c, -1.0);
    double p = fma(r, -1.0 / 6, 1.0 / 5);
    p = fma(r, p, -1.0 / 4);
    p = fma(r, p, 1.0 / 3);
    p = fma(r, p, -0.5);
    const double l1p = fma(r * r, p, r);
    return fma((double)k, 6.93147180559945286227e-01, lc + l1p);
}

// ln(x) for finite x > 0 (also subnormal); NaN propagates; x == 0 is not special-cased (the
// sweep has already turned such an entry into NaN through x / wth).  x = 2^k (1+f) with
// sqrt(1/2) <= 1+f < sqrt(2), s = f/(2+f), ln(1+f) = 2s + s*R(s^2) with the classical degree-7
// minimax R; < 1 ulp.  Branch-free so the sweep's inner loop stays one basic block.
__host__ __device__ __forceinline__ double dev_log(double x)
{
    int k;
    double m = sp_frexp(x, &k);                           // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    const double s = dev_div(f, d);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double Rp = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + Rp) + dk * ln2_lo)) - f);
}

// psi(x) and ln Gamma(x) together, x > 0.  D(x) ~ x^10 overflows fp64 past x ~ 6e30, so from x = 1e25 on the shift is
// left out (there the asymptotic series at y = x is exact to the last bit by itself; the engine's arguments are a shape plus
// a count sum, far below -- the guard makes the routine total, it is two selects).
// Both use the same upward shift by 10,
//     psi(x)     = psi(x+10)      - D'(x)/D(x)
//     lnGamma(x) = lnGamma(x+10)  - ln D(x),          D(x) = x (x+1) ... (x+9),
// then the Stirling / Bernoulli asymptotic series at y = x+10 >= 10, whose first omitted terms
// are < 5e-17.  One division, two logs, no loop, no branch.
//   |psi error|     <= ~2e-15 * max(1, |psi|)
//   |lnGamma error| <= ~1e-14 * max(1, |lnGamma|)   (two ~17.5-sized terms cancel near x = 1, 2)
__host__ __device__ __forceinline__ void dev_psi_lgamma(double x, double *psi, double *lgam)
{
    double D = x, Dp = 1.0;                               // D and dD/dx, built factor by factor
#pragma unroll
    for (int i = 1; i < 10; i++) {
        const double t = x + (double)i;
        Dp = fma(Dp, t, D);
        D = D * t;
    }
    const bool big = x > 1e25;                            // no shift: D would overflow from ~6e30 on
    const double y = big ? x : x + 10.0;
    const double ly = dev_log(y);
    const double yi = sp_rcp(y), y2 = yi * yi;
    // psi(y) = ln y - 1/(2y) - sum B_2k / (2k y^2k)
    double sp = 1.0 / 12;
    sp = fma(-y2, sp, 691.0 / 32760);
    sp = fma(-y2, sp, 1.0 / 132);
    sp = fma(-y2, sp, 1.0 / 240);
    sp = fma(-y2, sp, 1.0 / 252);
    sp = fma(-y2, sp, 1.0 / 120);
    sp = fma(-y2, sp, 1.0 / 12);
    *psi = (ly - 0.5 * yi - y2 * sp) - (big ? 0.0 : dev_div(Dp, D));
    // lnGamma(y) = (y - 1/2) ln y - y + ln(2 pi)/2 + sum B_2k / (2k (2k-1) y^(2k-1))
    double sg = 1.0 / 156;
    sg = fma(-y2, sg, 691.0 / 360360);
    sg = fma(-y2, sg, 1.0 / 1188);
    sg = fma(-y2, sg, 1.0 / 1680);
    sg = fma(-y2, sg, 1.0 / 1260);
    sg = fma(-y2, sg, 1.0 / 360);
    sg = fma(-y2, sg, 1.0 / 12);
    const double half_log_2pi = 0.91893853320467274178;
    *lgam = (((y - 0.5) * ly - y) + half_log_2pi + yi * sg) - (big ? 0.0 : dev_log(D));
}

}  // namespace vbnmf
