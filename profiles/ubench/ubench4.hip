// throughput of the non-FMA fp64 instructions the sweep uses
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters, double a)
{
    double v[10]; unsigned u[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { v[i] = 1.0 + threadIdx.x * 1e-3 + i; u[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            if (MODE == 0) v[i] = __builtin_amdgcn_rcp(v[i]);
            else if (MODE == 1) v[i] = __builtin_amdgcn_frexp_mant(v[i]) + 1.0;
            else if (MODE == 2) { u[i] += (unsigned)__builtin_amdgcn_frexp_exp(v[i]); }
            else if (MODE == 3) v[i] = (double)u[i] + v[i];
            else if (MODE == 4) v[i] = (double)__builtin_amdgcn_rcpf((float)v[i]);
            else if (MODE == 5) v[i] = __builtin_amdgcn_rsq(v[i]);
            else v[i] = fma(v[i], a, 1e-9);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s += v[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
#define RUN(M) timeit([&] { hipLaunchKernelGGL((k<M>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001); })
int main() {
    double *out; (void)hipMalloc(&out, 256 * 1024 * sizeof(double));
    const int iters = 5000;
    for (int nt : {256, 768}) {
        double wps = nt / 256.0, sc = 1e6 / (iters * 10.0 * wps);
        printf("nt=%4d: ns/op/SIMD  rcp_f64 %.2f | frexp_mant+add %.2f | frexp_exp+iadd %.2f | cvt_u32+add %.2f | cvt+rcp_f32+cvt %.2f | rsq_f64 %.2f | fma %.2f\n", nt,
               RUN(0) * sc, RUN(1) * sc, RUN(2) * sc, RUN(3) * sc, RUN(4) * sc, RUN(5) * sc, RUN(6) * sc);
    }
    return 0;
}
