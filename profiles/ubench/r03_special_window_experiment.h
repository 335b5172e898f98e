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


// ---- the window [1, 2] of psi and ln Gamma (round 3) ---------------------------------------------------------------
// Around their zeros (psi: z0 = 1.4616..., lnGamma: 1 and 2) the shift-by-10 forms below subtract two numbers of size 2.4-3
// (psi) or 17.5 (lnGamma) and are left with an ABSOLUTE error of a few ulp of those: 1.45e-15 for psi where |psi| < 0.6,
// and no relative accuracy for lnGamma near its zeros.  In the window both are written around their zeros,
//     psi(z) = (z - z0) g(z),      lnGamma(z) = (z - 1)(z - 2) h(z),      g, h polynomials of degree 20 in t = z - 1.5
// (interpolated at Chebyshev nodes in 60-digit arithmetic, profiles/ubench/gen_special_poly.py; the nearest singularity
// of g and h is the pole at 0, so the fit converges like 5.8^-n and degree 20 sits below the rounding of the Horner
// chain): |psi error| <= 1.2e-16 absolute, lnGamma to 3.5e-16 RELATIVE on [1, 2].  z0 is split so that z - z0hi is exact.
// coefficients, highest degree first: g (21), then h (21).  Kept in memory (host array / __constant__ device array that is
// NOT const-qualified, so the compiler cannot fold it back into 42 64-bit literals): as literals they took 84 registers of
// the update kernel and spilled 56 (k_update 26.8 -> 51.1 us); read through the scalar cache they cost no VGPR.
#define VBNMF_WINDOW_COEFFS { 0.0002522590506964885, -0.0003783907897248858, 0.00023650171719339814, -0.00035476352053371977, 0.0007184107969046378, -0.0010776863967234226, 0.0015580749137213761, -0.002337552647103118, 0.0035187504361950134, -0.005280903066090605, 0.007926972735646359, -0.011908146513976738, 0.017907250563619775, -0.026975800029253667, 0.0407608338843387, -0.06192213319072707, 0.09498872445354183, -0.14840492305485808, 0.24054248424078478, -0.4236274212814573, 0.9510558760318328, \
      1.1820969445467145e-05, -1.8527803997032962e-05, 1.3584275459130943e-05, -2.148770252062576e-05, 4.281642862794779e-05, -6.793733027397047e-05, 0.00010541715222185904, -0.0001685717945035172, 0.00027130273545114005, -0.00043803516132580534, 0.0007115202752051298, -0.001164411377219605, 0.0019229162321227923, -0.0032120742206072908, 0.005446457842219553, -0.009425622444829649, 0.01679709863121395, -0.03130848750105327, 0.06291140107456485, -0.14595989591430594, 0.4831289505409809 }
static const double kWindowHost[42] = VBNMF_WINDOW_COEFFS;
__device__ __constant__ double kWindowDev[42] = VBNMF_WINDOW_COEFFS;
#undef VBNMF_WINDOW_COEFFS
__host__ __device__ __forceinline__ void psi_lgamma_window(double z, double *psi, double *lgam)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double *c = kWindowDev;
#else
    const double *c = kWindowHost;
#endif
    const double t = z - 1.5;
    double g = c[0], h = c[21];
#pragma unroll
    for (int i = 1; i < 21; i++) {
        g = fma(g, t, c[i]);
        h = fma(h, t, c[21 + i]);
    }
    *psi = ((z - 1.4616321446374059) - 3.309564689176888e-10) * g;
    *lgam = (z - 1.0) * (z - 2.0) * h;
}

// psi(x) and ln Gamma(x) together, x > 0 (x up to ~1e25; the engine's arguments are count sums).
// Both use the same upward shift by 10,
//     psi(x)     = psi(x+10)      - D'(x)/D(x)
//     lnGamma(x) = lnGamma(x+10)  - ln D(x),          D(x) = x (x+1) ... (x+9),
// then the Stirling / Bernoulli asymptotic series at y = x+10 >= 10, whose first omitted terms
// are < 5e-17.  One division, two logs, no loop.  Arguments in (0, 4] take the window above instead (one branch: the
// update kernels are not bound by it), which is where this form's cancellation showed:
//   |psi error|     <= 1e-15 * max(1, |psi|)        (round 2, without the window: 1.45e-15 at 15 of 2 170 grid points)
//   |lnGamma error| <= ~1e-14 * max(1, |lnGamma|), and relative to |lnGamma| itself through its zeros x = 1, 2
__host__ __device__ __forceinline__ void dev_psi_lgamma(double x, double *psi, double *lgam)
{
    if (x > 0.0 && x <= 4.0) {
        // One or two steps of the recurrences psi(x+1) = psi(x) + 1/x, lnGamma(x+1) = lnGamma(x) + ln x into the window [1, 2].
        // Downwards (x > 2) the steps are exact (x - 1 is representable); upwards (x < 1) x + 1 rounds, and the part lost
        // -- zl, exact by the two-sum below -- re-enters through the derivatives: lnGamma'(z) = psi(z), psi'(z) ~ 1/z + 1/(2 z^2).
        double z = x, pc = 0.0, lc = 0.0, zl = 0.0;
        if (z < 1.0) {
            pc = -dev_div(1.0, z); lc = -dev_log(z);
            const double zh = z + 1.0;
            zl = z - (zh - 1.0);
            z = zh;
        } else {
            if (z > 3.0) { z -= 1.0; pc += dev_div(1.0, z); lc += dev_log(z); }
            if (z > 2.0) { z -= 1.0; pc += dev_div(1.0, z); lc += dev_log(z); }
        }
        double pw, lw;
        psi_lgamma_window(z, &pw, &lw);
        const double zi = dev_div(1.0, z);
        *psi = fma(zl, fma(0.5 * zi, zi, zi), pw) + pc;
        *lgam = fma(zl, pw, lw) + lc;
        return;
    }
    double D = x, Dp = 1.0;                               // D and dD/dx, built factor by factor
#pragma unroll
    for (int i = 1; i < 10; i++) {
        const double t = x + (double)i;
        Dp = fma(Dp, t, D);
        D = D * t;
    }
    const double y = x + 10.0;
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
    *psi = (ly - 0.5 * yi - y2 * sp) - dev_div(Dp, D);
    // lnGamma(y) = (y - 1/2) ln y - y + ln(2 pi)/2 + sum B_2k / (2k (2k-1) y^(2k-1))
    double sg = 1.0 / 156;
    sg = fma(-y2, sg, 691.0 / 360360);
    sg = fma(-y2, sg, 1.0 / 1188);
    sg = fma(-y2, sg, 1.0 / 1680);
    sg = fma(-y2, sg, 1.0 / 1260);
    sg = fma(-y2, sg, 1.0 / 360);
    sg = fma(-y2, sg, 1.0 / 12);
    const double half_log_2pi = 0.91893853320467274178;
    *lgam = (((y - 0.5) * ly - y) + half_log_2pi + yi * sg) - dev_log(D);
}

}  // namespace vbnmf
