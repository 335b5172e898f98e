// per-entry arithmetic of the sweep from registers only: how close to the fp64 issue roof can it get?
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dv(double x, double w) {
    double rc = __builtin_amdgcn_rcp(w);
    rc = fma(fma(-w, rc, 1.0), rc, rc);
    const double q = x * rc;
    return fma(fma(-w, q, x), rc, q);
}
template <int R> struct Regs { double F[R], acc[R]; };
template <int R> __device__ __forceinline__ void entry(Regs<R> &S, const double (&g)[R], double x) {
    double w0 = S.F[0] * g[0], w1 = S.F[1] * g[1];
#pragma unroll
    for (int k = 2; k < R; k += 2) { w0 = fma(S.F[k], g[k], w0); w1 = fma(S.F[k + 1], g[k + 1], w1); }
    const double q = dv(x, w0 + w1);
#pragma unroll
    for (int k = 0; k < R; k++) S.acc[k] = fma(q, g[k], S.acc[k]);
}
template <int R, int MODE>
__global__ void k(double *out, int iters, double seed)
{
    Regs<R> S;
    double g[R], h[R];
#pragma unroll
    for (int k = 0; k < R; k++) { S.F[k] = 1.0 + 0.01 * k + threadIdx.x * 1e-6; S.acc[k] = 0; g[k] = 0.5 + 0.02 * k; h[k] = 0.7 + 0.01 * k; }
    double x = seed;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            entry<R>(S, g, x); __builtin_amdgcn_sched_barrier(0);
            entry<R>(S, h, x + 1.0); __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 1) {            // compiler free to interleave the two entries
            entry<R>(S, g, x);
            entry<R>(S, h, x + 1.0);
        } else {                           // two independent accumulator sets: fully independent entries
            Regs<R> &T = S;
            double w0 = T.F[0] * g[0], w1 = T.F[1] * g[1], v0 = T.F[0] * h[0], v1 = T.F[1] * h[1];
#pragma unroll
            for (int k = 2; k < R; k += 2) { w0 = fma(T.F[k], g[k], w0); v0 = fma(T.F[k], h[k], v0); w1 = fma(T.F[k + 1], g[k + 1], w1); v1 = fma(T.F[k + 1], h[k + 1], v1); }
            const double wa = w0 + w1, wb = v0 + v1;
            double ra = __builtin_amdgcn_rcp(wa), rb = __builtin_amdgcn_rcp(wb);
            ra = fma(fma(-wa, ra, 1.0), ra, ra); rb = fma(fma(-wb, rb, 1.0), rb, rb);
            double qa = x * ra, qb = (x + 1.0) * rb;
            qa = fma(fma(-wa, qa, x), ra, qa); qb = fma(fma(-wb, qb, x + 1.0), rb, qb);
#pragma unroll
            for (int k = 0; k < R; k++) T.acc[k] = fma(qa, g[k], fma(qb, h[k], T.acc[k]));
        }
        g[0] += 1e-9; h[3] += 1e-9; x += 1e-6;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; k++) s += S.acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    double *out; (void)hipMalloc(&out, 256 * 1024 * sizeof(double));
    const int iters = 20000;
    for (int nt : {256, 512, 768, 1024}) {
        float a = timeit([&] { hipLaunchKernelGGL((k<10, 0>), dim3(256), dim3(nt), 0, 0, out, iters, 3.0); });
        float b = timeit([&] { hipLaunchKernelGGL((k<10, 1>), dim3(256), dim3(nt), 0, 0, out, iters, 3.0); });
        float c = timeit([&] { hipLaunchKernelGGL((k<10, 2>), dim3(256), dim3(nt), 0, 0, out, iters, 3.0); });
        double wps = nt / 256.0;
        printf("nt=%4d waves/SIMD=%.0f : ns per entry per SIMD-slot: fenced %.1f  free %.1f  paired %.1f\n", nt, wps,
               a * 1e6 / (2.0 * iters * wps), b * 1e6 / (2.0 * iters * wps), c * 1e6 / (2.0 * iters * wps));
    }
    return 0;
}
