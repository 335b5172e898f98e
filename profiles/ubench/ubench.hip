// microbenchmarks: fp64 VALU issue rate and ds_read_b128 throughput on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_fma(double *out, int iters, double a, double b)
{
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) v[i] = fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mix: per iteration 5 ds_read_b128 (random rows or same row) + NF dependent-free fmas
template <int NF>
__global__ void k_lds(double *out, int iters, int stride_mode, double a)
{
    extern __shared__ double2 lds[];
    for (int t = threadIdx.x; t < 9600; t += blockDim.x) lds[t] = make_double2(t * 1e-4, 1.0);
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u) >> 8;
    double acc[10] = {0};
    for (int it = 0; it < iters; it++) {
        unsigned row = stride_mode == 0 ? 0u : (stride_mode == 1 ? (idx % 1900u) : ((threadIdx.x & 63u) + (it & 7) * 64u));
        idx = idx * 1664525u + 1013904223u;
        const double2 *g = lds + row * 5;
        double2 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
        double w = 0;
        w = fma(g0.x, a, w); w = fma(g0.y, a, w); w = fma(g1.x, a, w); w = fma(g1.y, a, w); w = fma(g2.x, a, w);
        w = fma(g2.y, a, w); w = fma(g3.x, a, w); w = fma(g3.y, a, w); w = fma(g4.x, a, w); w = fma(g4.y, a, w);
#pragma unroll
        for (int i = 0; i < NF; i++) acc[i % 10] = fma(w, a, acc[i % 10]);
    }
    double s = 0;
    for (int i = 0; i < 10; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    double *out; CK(hipMalloc(&out, 256 * 1024 * sizeof(double)));
    const int iters = 20000;
    for (int nt : {256, 512, 768, 1024}) {
        float m1 = timeit([&] { hipLaunchKernelGGL((k_fma<1>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        float m4 = timeit([&] { hipLaunchKernelGGL((k_fma<4>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        float m10 = timeit([&] { hipLaunchKernelGGL((k_fma<10>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        // cycles per wave-instr per SIMD = time * clk / (iters*ILP*waves_per_simd)
        double wps = nt / 256.0;
        printf("fma nt=%4d waves/SIMD=%.0f : ILP1 %.2f ns/instr/SIMD  ILP4 %.2f  ILP10 %.2f\n", nt, wps,
               m1 * 1e6 / (iters * 1 * wps), m4 * 1e6 / (iters * 4 * wps), m10 * 1e6 / (iters * 10 * wps));
    }
    hipFuncSetAttribute((const void *)k_lds<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_lds<40>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nt : {256, 768}) for (int mode : {0, 1, 2}) {
        float a = timeit([&] { hipLaunchKernelGGL((k_lds<10>), dim3(256), dim3(nt), 153600, 0, out, 4000, mode, 1.0000001); });
        float b = timeit([&] { hipLaunchKernelGGL((k_lds<40>), dim3(256), dim3(nt), 153600, 0, out, 4000, mode, 1.0000001); });
        printf("lds nt=%4d mode=%d (0 bcast,1 random,2 linear): 20 fma + 5 b128 per iter: %.1f ns/iter/wave-slot ; 50 fma: %.1f\n", nt, mode,
               a * 1e6 / 4000 / (nt / 256.0), b * 1e6 / 4000 / (nt / 256.0));
    }
    return 0;
}
