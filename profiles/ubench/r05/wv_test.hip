#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_set(unsigned *f, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence(); __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } }
__global__ void k_busy(double *p, int n) { double s = p[threadIdx.x]; for (int i = 0; i < n; i++) s = s * 1.0000001 + 1e-9; p[threadIdx.x] = s; }
int main() {
    int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue = %d\n", can);
    unsigned *flag; hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    double *buf; hipMalloc(&buf, 8 * 1024); hipMemset(buf, 0, 8 * 1024);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e0, e1, ev; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 20; rep++) {
            hipMemsetAsync(flag, 0, 4, a); hipStreamSynchronize(a);
            hipEventRecord(e0, b);
            // stream a: busy kernel, then signal; stream b: wait, then a kernel; measure b's total
            hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, a, buf, 20000);
            if (mode == 0) { hipEventRecord(ev, a); hipStreamWaitEvent(b, ev, 0); }
            else { hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, a, flag, 1u); hipError_t r = hipStreamWaitValue32(b, flag, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu); if (r != hipSuccess) { printf("WaitValue failed: %s\n", hipGetErrorString(r)); return 1; } }
            hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, b, buf + 512, 10);
            hipEventRecord(e1, b);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s: stream b done %.1f us after its start (busy kernel on a ~ fixed)\n", mode == 0 ? "event record + stream wait event" : "flag kernel + hipStreamWaitValue32", best * 1e3);
    }
    return 0;
}
