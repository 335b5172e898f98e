// special.h -- fp64 digamma and log-gamma for gfx950 device code.
//
// The reference evaluates these through GSL (gsl_sf_psi, gsl_sf_lngamma; reference
// src/vbnmf_update.cpp:59,63,82,85,87,89).  HIP has no digamma, and the engine only needs
// positive arguments (alw = aw + sw >= aw > 0), so both are written out here:
// upward recurrence to x >= 10, then the Stirling / Bernoulli asymptotic series, whose
// first omitted term is < 1e-16 there.
#pragma once
#include <hip/hip_runtime.h>

namespace vbnmf {

// psi(x), x > 0.  |error| <= ~1e-15 * max(1, |psi|) (checked against mpmath in tests).
__device__ __forceinline__ double dev_digamma(double x)
{
    if (!(x > 0.0)) return __builtin_nan("");
    double s = 0.0;
    while (x < 10.0) { s -= 1.0 / x; x += 1.0; }
    const double xi = 1.0 / x, y = xi * xi;
    double ser = 1.0 / 12;                       // B_14/14 = 1/12
    ser = 691.0 / 32760 - y * ser;
    ser = 1.0 / 132 - y * ser;
    ser = 1.0 / 240 - y * ser;
    ser = 1.0 / 252 - y * ser;
    ser = 1.0 / 120 - y * ser;
    ser = 1.0 / 12 - y * ser;
    return s + log(x) - 0.5 * xi - y * ser;
}

// ln Gamma(x), x > 0.  |error| <= ~4e-15 * max(1, |lnGamma|).
__device__ __forceinline__ double dev_lgamma(double x)
{
    if (!(x > 0.0)) return (x == 0.0) ? __builtin_inf() : __builtin_nan("");
    double p = 1.0;
    while (x < 10.0) { p *= x; x += 1.0; }
    const double xi = 1.0 / x, y = xi * xi;
    double ser = 1.0 / 156;
    ser = 691.0 / 360360 - y * ser;
    ser = 1.0 / 1188 - y * ser;
    ser = 1.0 / 1680 - y * ser;
    ser = 1.0 / 1260 - y * ser;
    ser = 1.0 / 360 - y * ser;
    ser = 1.0 / 12 - y * ser;
    const double half_log_2pi = 0.91893853320467274178;
    double st = (x - 0.5) * log(x) - x + half_log_2pi + xi * ser;
    return st - log(p);
}

}  // namespace vbnmf
