// same-address atomic tickets under the sweep's contention: G workgroups x 12 waves per counter, each wave takes T tickets with ~W us of work between
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned queue_issue(unsigned *q)
{
    unsigned ret; unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(ret), "=&s"(save) : "v"(0u), "v"(1u), "s"(q) : "memory");
    return ret;
}
__device__ __forceinline__ int queue_take(unsigned ret)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ret) :: "memory");
    return __builtin_amdgcn_readfirstlane((int)ret);
}
__global__ __launch_bounds__(768) void k(unsigned *queues, int wg_per_q, int tickets, int work, unsigned long long *lat, double *sink)
{
    const int wg = (int)(blockIdx.x % 8) * (gridDim.x / 8) + (int)(blockIdx.x / 8);
    unsigned *q = queues + (wg / wg_per_q) * 64;           // counters 256 bytes apart
    double acc = threadIdx.x;
    unsigned long long tl = 0;
    for (int t = 0; t < tickets; t++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const int i = queue_take(queue_issue(q));
        tl += __builtin_amdgcn_s_memrealtime() - t0;
        for (int w = 0; w < work; w++) acc = fma(acc, 1.0000001, (double)i);
    }
    if ((threadIdx.x & 63) == 0) lat[blockIdx.x * 12 + (threadIdx.x >> 6)] = tl;
    if (acc == 12345.678) *sink = acc;
}
int main()
{
    unsigned *queues; unsigned long long *lat; double *sink;
    hipMalloc(&queues, 64 * 64 * 4); hipMalloc(&lat, 256 * 12 * 8); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][3] = {{10, 3, 4000}, {10, 3, 0}, {1, 3, 4000}, {32, 3, 4000}, {256, 3, 4000}, {10, 30, 400}};
    for (auto &c : cfg) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(queues, 0, 64 * 64 * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, queues, c[0], c[1], c[2], lat, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(256 * 12);
            hipMemcpy(h.data(), lat, h.size() * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0; for (auto v : h) { s += v; if (v > mx) mx = v; }
            if (rep == 2) printf("wg/queue %3d tickets/wave %2d work %4d: kernel %.1f us; ticket latency mean %.2f us, worst wave mean %.2f us\n", c[0], c[1], c[2], ms * 1e3, s / h.size() / c[1] / 100.0, mx / c[1] / 100.0);
        }
    }
    return 0;
}
