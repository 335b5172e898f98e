// fp64 FMA issue rate by operand kind: v = fma(v, s, s) / fma(v, v, s) / fma(v, v, v)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters, double a, double b)
{
    double v[10], u[10], t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { v[i] = threadIdx.x * 1e-3 + i; u[i] = 1.0 + 1e-7 * (threadIdx.x + i); t[i] = 1e-9 * (i + 1 + threadIdx.x); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            if (MODE == 0) v[i] = fma(v[i], a, b);
            else if (MODE == 1) v[i] = fma(v[i], u[i], b);
            else if (MODE == 2) v[i] = fma(v[i], u[i], t[i]);
            else if (MODE == 3) v[i] = fma(u[i], t[i], v[i]);          // accumulate form: dst == src2
            else if (MODE == 4) v[i] = v[i] * u[i];
            else if (MODE == 5) v[i] = v[i] + u[i];
            else v[i] = fma(u[(i + 1) % 10], t[(i + 3) % 10], v[i]);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s += v[i] + u[i] + t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
#define RUN(M) timeit([&] { hipLaunchKernelGGL((k<M>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); })
int main() {
    double *out; (void)hipMalloc(&out, 256 * 1024 * sizeof(double));
    const int iters = 20000;
    for (int nt : {256, 768}) {
        double wps = nt / 256.0, sc = 1e6 / (iters * 10.0 * wps);
        printf("nt=%4d: ns/instr/SIMD  fma(v,s,s) %.2f | fma(v,v,s) %.2f | fma(v,v,v) %.2f | acc=fma(v,v,acc) %.2f | mul(v,v) %.2f | add(v,v) %.2f | fma(v',v'',acc) %.2f\n", nt,
               RUN(0) * sc, RUN(1) * sc, RUN(2) * sc, RUN(3) * sc, RUN(4) * sc, RUN(5) * sc, RUN(6) * sc);
    }
    return 0;
}
