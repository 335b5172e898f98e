// microbenchmark: can ds_add_f64 (LDS fp64 atomics, no return) carry the cell-side scatter of a one-pass sweep?
// per "entry": 5 ds_read_b128 (one gathered row of R=10) + NF fp64 FMAs [+ 10 ds_add_f64 to a random row]
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NF, int NATOM>
__global__ __launch_bounds__(768) void k_mix(double *out, int iters, double a)
{
    extern __shared__ double2 lds[];
    for (int t = threadIdx.x; t < 10000; t += blockDim.x) lds[t] = make_double2(t * 1e-4, 1.0);
    __syncthreads();
    double *accblk = reinterpret_cast<double *>(lds + 5000);   // second half: 1000 rows x 10 doubles
    unsigned idx = (threadIdx.x * 2654435761u + blockIdx.x * 977u) >> 8;
    double acc[10] = {0};
    for (int it = 0; it < iters; it++) {
        const unsigned row = idx % 1000u;
        idx = idx * 1664525u + 1013904223u;
        const double2 *g = lds + row * 5;
        double2 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
        double w = 0;
        w = fma(g0.x, a, w); w = fma(g0.y, a, w); w = fma(g1.x, a, w); w = fma(g1.y, a, w); w = fma(g2.x, a, w);
        w = fma(g2.y, a, w); w = fma(g3.x, a, w); w = fma(g3.y, a, w); w = fma(g4.x, a, w); w = fma(g4.y, a, w);
#pragma unroll
        for (int i = 0; i < NF; i++) acc[i % 10] = fma(w, a, acc[i % 10]);
        if (NATOM) {
            double *dst = accblk + row * 10;
#pragma unroll
            for (int k = 0; k < NATOM; k++)
                __hip_atomic_fetch_add(dst + k, w * (k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    double s = 0;
    for (int i = 0; i < 10; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + accblk[threadIdx.x];
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
    const int iters = 4000;
    const size_t lds = 160000;
#define RUN(NF, NA) { CK(hipFuncSetAttribute((const void *)k_mix<NF, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        float ms = timeit([&] { k_mix<NF, NA><<<256, 768, lds>>>(out, iters, 1.0000001); }); \
        printf("NF=%d NATOM=%d : %.3f ms, %.2f ns per entry per wave (3 waves/SIMD => x3 per SIMD-slot)\n", NF, NA, ms, ms * 1e6 / iters); }
    RUN(20, 0) RUN(20, 10) RUN(37, 0) RUN(37, 10) RUN(47, 0) RUN(47, 10) RUN(0, 10) RUN(0, 0)
    CK(hipDeviceSynchronize());
    return 0;
}
