// mfma_f64.hip -- fp64 matrix-core microbenchmark for row n1 of the scope table (north_star: "MFMA used only for the
// dense W^T (X/WH) and (X/WH) H^T contractions at larger ranks"; reference GEMMs: src/vbnmf_update.cpp:33,35,36).
//
//   1. issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 (independent accumulators, 1..4 waves / SIMD)
//   2. fp64 VALU FMAs issued BETWEEN those MFMAs: do the two pipes overlap?
//   3. fragment layout check of the 16x16x4 form against a host product
//   4. the instruction mix of ONE dense 16 genes x 16 cells tile of the VB sweep done with MFMA (rank <= 8):
//        wth = LW LH            2 MFMA     q  = X / wth    (4 per lane)     sh += q^T LW   4 MFMA   (genes {4k+v}, no permute)
//        wthT = LH^T LW^T       2 MFMA     qT = X^T / wthT (4 per lane)     sw += q LH^T   4 MFMA
//        + sum x log(wth) on one copy (4 logs per lane), all on in-register data: cycles per tile = an UPPER bound of
//      what a dense-X kernel built this way can reach; compared with the entries/s the VALU sweep measures on C2.
//
// hipcc -O3 --offload-arch=gfx950 -o mfma_f64 mfma_f64.hip ; run on the GPU box; output kept in profiles/r02_mfma_f64.txt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int NFMA>
__global__ void k_mfma16(double *out, int iters, double a, double b)
{
    v4d acc[NACC];
    double f[NFMA > 0 ? NFMA : 1];
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
    for (int i = 0; i < NFMA; i++) f[i] = threadIdx.x * 1e-3 + i;
    const double x = a + threadIdx.x * 1e-6, y = b - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NFMA / NACC; q++) f[i * (NFMA / NACC) + q] = fma(f[i * (NFMA / NACC) + q], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < NFMA; i++) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k_mfma4(double *out, int iters, double a, double b)
{
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = 0;
    const double x = a + threadIdx.x * 1e-6, y = b - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout check: D = A (16x4) B (4x16) with A[i][k] = 1 + i + 100 k, B[k][j] = 2 + j + 1000 k
__global__ void k_layout(double *out)
{
    const int l = threadIdx.x;
    const double a = 1.0 + (l % 16) + 100.0 * (l / 16);           // A[i = l % 16][k = l / 16]
    const double b = 2.0 + (l % 16) + 1000.0 * (l / 16);          // B[k = l / 16][j = l % 16]
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, (v4d){0, 0, 0, 0}, 0, 0, 0);
    out[l * 4 + 0] = d.x; out[l * 4 + 1] = d.y; out[l * 4 + 2] = d.z; out[l * 4 + 3] = d.w;
}

__device__ __forceinline__ double rcp_fast(double w)
{
    const double rc = __builtin_amdgcn_rcp(w);
    return fma(fma(-w, rc, 1.0), rc, rc);
}
__device__ __forceinline__ double log_poly(double x)                 // the sweep's table-driven ln without the table read (same count)
{
    const int k = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);
    const double r = fma(m, 1.3, -1.0);
    double p = fma(r, -1.0 / 6, 1.0 / 5);
    p = fma(r, p, -1.0 / 4); p = fma(r, p, 1.0 / 3); p = fma(r, p, -0.5);
    return fma((double)k, 0.6931471805599453, fma(r * r, p, r));
}

// one 16 x 16 tile per wave and iteration, operands in registers (the LDS / HBM side is left out: upper bound)
__global__ void k_tile(double *out, int iters, double seed)
{
    const int l = threadIdx.x & 63;
    double lw0 = seed + l * 1e-3, lw1 = seed * 0.5 + l * 2e-3, lh0 = seed * 0.25 + l * 1e-3, lh1 = seed * 0.125 + l * 3e-3;
    v4d x = {1.0, 2.0, 0.0, 1.0};
    v4d sh = {0, 0, 0, 0}, sw = {0, 0, 0, 0};
    double lsum = 0;
    for (int it = 0; it < iters; it++) {
        v4d w = __builtin_amdgcn_mfma_f64_16x16x4f64(lw0, lh0, (v4d){0, 0, 0, 0}, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(lw1, lh1, w, 0, 0, 0);
        v4d wt = __builtin_amdgcn_mfma_f64_16x16x4f64(lh0, lw0, (v4d){0, 0, 0, 0}, 0, 0, 0);
        wt = __builtin_amdgcn_mfma_f64_16x16x4f64(lh1, lw1, wt, 0, 0, 0);
        v4d q, qt;
        q.x = x.x * rcp_fast(w.x + 1.0); q.y = x.y * rcp_fast(w.y + 1.0); q.z = x.z * rcp_fast(w.z + 1.0); q.w = x.w * rcp_fast(w.w + 1.0);
        qt.x = x.x * rcp_fast(wt.x + 1.0); qt.y = x.y * rcp_fast(wt.y + 1.0); qt.z = x.z * rcp_fast(wt.z + 1.0); qt.w = x.w * rcp_fast(wt.w + 1.0);
        lsum = fma(x.x, log_poly(w.x + 1.0), lsum); lsum = fma(x.y, log_poly(w.y + 1.0), lsum);
        lsum = fma(x.z, log_poly(w.z + 1.0), lsum); lsum = fma(x.w, log_poly(w.w + 1.0), lsum);
        sh = __builtin_amdgcn_mfma_f64_16x16x4f64(q.x, lw0, sh, 0, 0, 0);
        sh = __builtin_amdgcn_mfma_f64_16x16x4f64(q.y, lw1, sh, 0, 0, 0);
        sh = __builtin_amdgcn_mfma_f64_16x16x4f64(q.z, lw0, sh, 0, 0, 0);
        sh = __builtin_amdgcn_mfma_f64_16x16x4f64(q.w, lw1, sh, 0, 0, 0);
        sw = __builtin_amdgcn_mfma_f64_16x16x4f64(qt.x, lh0, sw, 0, 0, 0);
        sw = __builtin_amdgcn_mfma_f64_16x16x4f64(qt.y, lh1, sw, 0, 0, 0);
        sw = __builtin_amdgcn_mfma_f64_16x16x4f64(qt.z, lh0, sw, 0, 0, 0);
        sw = __builtin_amdgcn_mfma_f64_16x16x4f64(qt.w, lh1, sw, 0, 0, 0);
        lw0 += 1e-9; lh0 += 1e-9;                                    // keep the compiler from hoisting the products
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sh.x + sh.y + sh.z + sh.w + sw.x + sw.y + sw.z + sw.w + lsum;
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
    double *out; CK(hipMalloc(&out, 256 * 1024 * 4 * sizeof(double)));
    const int iters = 20000;
    printf("== v_mfma_f64_16x16x4_f64 (2048 flop), 256 workgroups, independent accumulators\n");
    for (int nt : {256, 512, 768, 1024}) {
        const double wps = nt / 256.0;
        float m1 = timeit([&] { hipLaunchKernelGGL((k_mfma16<1, 0>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        float m4 = timeit([&] { hipLaunchKernelGGL((k_mfma16<4, 0>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        const double ns1 = m1 * 1e6 / (iters * 1 * wps), ns4 = m4 * 1e6 / (iters * 4 * wps);
        printf("nt=%4d waves/SIMD=%.0f : 1 acc %.2f ns/MFMA/SIMD ; 4 acc %.2f ns/MFMA/SIMD = %.1f TFLOP/s chip\n", nt, wps, ns1, ns4,
               2048.0 / ns4 * 1024 / 1e3);
    }
    printf("== fp64 VALU FMAs between the MFMAs (4 accumulators; NFMA independent FMAs per 4 MFMAs), nt = 768\n");
    {
        const double wps = 3.0;
        float a0 = timeit([&] { hipLaunchKernelGGL((k_mfma16<4, 0>), dim3(256), dim3(768), 0, 0, out, iters, 1.0000001, 1e-9); });
        float a16 = timeit([&] { hipLaunchKernelGGL((k_mfma16<4, 16>), dim3(256), dim3(768), 0, 0, out, iters, 1.0000001, 1e-9); });
        float a32 = timeit([&] { hipLaunchKernelGGL((k_mfma16<4, 32>), dim3(256), dim3(768), 0, 0, out, iters, 1.0000001, 1e-9); });
        float a64 = timeit([&] { hipLaunchKernelGGL((k_mfma16<4, 64>), dim3(256), dim3(768), 0, 0, out, iters, 1.0000001, 1e-9); });
        printf("per 4 MFMAs per wave-slot: +0 FMA %.1f ns | +16 FMA %.1f | +32 FMA %.1f | +64 FMA %.1f   (4 MFMA = 8192 flop, 64 FMA wave-instr = 8192 flop)\n",
               a0 * 1e6 / iters / wps, a16 * 1e6 / iters / wps, a32 * 1e6 / iters / wps, a64 * 1e6 / iters / wps);
    }
    printf("== v_mfma_f64_4x4x4_4b_f64 (512 flop)\n");
    for (int nt : {256, 768}) {
        const double wps = nt / 256.0;
        float m4 = timeit([&] { hipLaunchKernelGGL((k_mfma4<4>), dim3(256), dim3(nt), 0, 0, out, iters, 1.0000001, 1e-9); });
        const double ns4 = m4 * 1e6 / (iters * 4 * wps);
        printf("nt=%4d : %.2f ns/MFMA/SIMD = %.1f TFLOP/s chip\n", nt, ns4, 512.0 / ns4 * 1024 / 1e3);
    }
    printf("== fragment layout of the 16x16x4 form\n");
    {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
        std::vector<double> h(256);
        CK(hipMemcpy(h.data(), out, 256 * sizeof(double), hipMemcpyDeviceToHost));
        int bad_a = 0, bad_b = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
            auto ref = [&](int i, int j) { double s = 0; for (int k = 0; k < 4; k++) s += (1.0 + i + 100.0 * k) * (2.0 + j + 1000.0 * k); return s; };
            if (h[l * 4 + v] != ref(4 * (l / 16) + v, l % 16)) bad_a++;       // D[i = 4 (l/16) + v][j = l % 16]
            if (h[l * 4 + v] != ref((l / 16) + 4 * v, l % 16)) bad_b++;       // D[i = (l/16) + 4 v][j = l % 16]
        }
        printf("lane l, register v holds D[4 (l/16) + v][l %% 16]: %s ; D[(l/16) + 4 v][l %% 16]: %s\n", bad_a ? "no" : "YES", bad_b ? "no" : "YES");
    }
    printf("== one dense 16 x 16 tile of the VB sweep per wave and iteration (12 MFMA + 8 divisions + 4 logs per lane), operands in registers\n");
    for (int nt : {256, 512, 768, 1024}) {
        const double wps = nt / 256.0;
        float m = timeit([&] { hipLaunchKernelGGL(k_tile, dim3(256), dim3(nt), 0, 0, out, 4000, 0.37); });
        const double ns = m * 1e6 / (4000 * wps);
        printf("nt=%4d : %.1f ns per tile per SIMD = %.3f ns per matrix element per SIMD = %.1f G elements/s chip\n", nt, ns, ns / 256, 256.0 / ns * 1024);
    }
    return 0;
}
