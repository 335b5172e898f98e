// fake_rccl.cpp -- TEST INFRASTRUCTURE, not a product fallback: a stand-in for librccl that lets TWO (or more) processes
// on ONE GPU run the library's RCCL protocol (ccfindr_amd/csrc/comm.h opens it when VBNMF_RCCL_LIB names it).  Real RCCL
// refuses two ranks on a device and the test box has one GPU, so without this the code that only runs with nranks > 1 --
// vbnmf_comm_create, queue_vb_step's RCCL branch, the event ring, drive_loop's replicated queueing, the bounded waits
// with a dead peer -- would first execute on the driver's 8-GPU node.  The product keeps real RCCL over xGMI.
//
// Exports the eight symbols comm.h binds, with RCCL's signatures.  A collective is STREAM-ORDERED like the real one:
//   hipMemcpyAsync(device -> this rank's slot in a shared-memory segment)        on the caller's stream
//   hipLaunchHostFunc: publish "slot k ready", wait for every rank's, sum the slots IN RANK ORDER into a private
//                      pinned buffer, publish "read k", wait for every rank's (then the slots may be overwritten)
//   hipMemcpyAsync(private buffer -> device)
// so kernels queued behind it on the stream see the reduced data, kernels on other streams run beside it, and a rank
// whose peer never arrives stays on hipErrorNotReady -- exactly what the library's timeouts must handle.  The wait inside
// the host function gives up after FAKE_RCCL_TIMEOUT_S (default 60) so that a process can still exit.
//
// FAKE_RCCL_KERNEL=1 (round 5): the collective is carried by KERNELS, like the real one -- a host function occupies no CU
// and no LDS, so it could not show whether an all-reduce RUNS beside the persistent cell-side sweep on the CUs
// VBNMF_COMM_CUS leaves free.  Every rank owns an exchange buffer in device memory, opened by its peers through HIP IPC
// (hipIpcGetMemHandle / hipIpcOpenMemHandle; the handles travel through the shared segment): [flags | slot 0 | slot 1].
//   k_copy_in  (a few blocks)            waits until every peer has finished reading this parity's slot two collectives ago,
//                                        copies `send` into the rank's slot, publishes ready = k + 1 (system-scope atomic)
//   k_reduce   (FAKE_RCCL_BLOCKS x 512   spins (bounded) until every rank's ready >= k + 1, sums the peers' slots IN RANK
//               threads, 4 KB of LDS)    ORDER straight into `recv`, publishes done = k + 1
// -- RCCL's launch shape (tens of blocks, a few KB of LDS each, spinning on peers' flags), the same stream order.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = (size_t)64 << 20;          // per rank; the segment is sparse until written
constexpr size_t kHeaderBytes = 4096;

struct Header {
    std::atomic<uint32_t> arrived;                       // ranks that mapped the segment
    std::atomic<uint32_t> left;                          // ranks that destroyed their communicator
    std::atomic<uint64_t> ready[kMaxRanks];              // ops whose input slot is complete, per rank
    std::atomic<uint64_t> read[kMaxRanks];               // ops whose inputs this rank has finished reading
    std::atomic<uint32_t> ipc_ready[kMaxRanks];          // kernel-carried form: this rank's IPC handle is in place
    std::atomic<uint32_t> ipc_opened;                    // ... ranks that have opened every peer's buffer
    hipIpcMemHandle_t ipc[kMaxRanks];
};
static_assert(sizeof(Header) <= kHeaderBytes, "header too large");

// ---- kernel-carried form -----------------------------------------------------------------------------------------
constexpr size_t kDevSlotBytes = (size_t)8 << 20;        // per parity
constexpr size_t kDevHeaderBytes = 4096;
struct DevHeader {                                       // at the start of a rank's exchange buffer; written by its owner only
    unsigned long long ready;                            // collectives whose input slot is complete
    unsigned long long done;                             // collectives whose inputs this rank has finished reading
    unsigned int arrive_in, arrive_red;                  // block counters of the two kernels (owner-local)
    unsigned int broken;
};
struct PeerTable { char *buf[kMaxRanks]; };

__device__ inline const DevHeader *hdr(const PeerTable &P, int r) { return reinterpret_cast<const DevHeader *>(P.buf[r]); }
__device__ inline double *dslot(const PeerTable &P, int r, unsigned long long k)
{
    return reinterpret_cast<double *>(P.buf[r] + kDevHeaderBytes + (size_t)(k & 1) * kDevSlotBytes);
}
// thread 0 of the block: wait until every rank's flag (ready: which = 0, done: which = 1) has reached `want`; bounded
__device__ inline bool wait_flags(const PeerTable &P, int nranks, int which, unsigned long long want, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
        for (int r = 0; r < nranks; r++) {
            const unsigned long long *f = which ? &hdr(P, r)->done : &hdr(P, r)->ready;
            ok = ok && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want;
        }
        if (ok) return true;
        if (wall_clock64() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ __launch_bounds__(512) void k_copy_in(PeerTable P, int nranks, int rank, const double *__restrict__ send, size_t count,
                                                 unsigned long long k, unsigned long long ticks, int vec)
{
    __shared__ int ok;
    __shared__ double lds_pad[512];                      // 4 KB, like k_reduce: RCCL's kernels hold LDS, so they only fit CUs whose LDS
                                                         // the persistent sweep has left free (without it the block lands on ANY CU and is
                                                         // starved by the sweep's older waves: 160 us instead of 26, r05_c5_overlap.txt)
    lds_pad[threadIdx.x] = (double)threadIdx.x;
    DevHeader *me = reinterpret_cast<DevHeader *>(P.buf[rank]);
    if (threadIdx.x == 0) ok = (k < 2) ? 1 : (wait_flags(P, nranks, 1, k - 1, ticks) ? 1 : 0);     // this parity's slot is free again
    // (send / recv / the slots are 16-byte aligned: hipMalloc'd buffers and offsets of whole doubles in pairs -- the library's
    // reduce buffers start on allocation boundaries; an odd offset would need the scalar path)
    __syncthreads();
    if (ok) {
        // 16-byte accesses, eight of them in flight per thread (a one-element-per-trip copy is a chain of memory latencies:
        // 80 us for 4.8 MB alone on the chip, 200 us beside a sweep -- the first version of this kernel)
        double *dst = dslot(P, rank, k);
        if (!vec) {                                      // (a buffer that does not start on 16 bytes: element by element)
            for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < count; i += (size_t)gridDim.x * 512) dst[i] = send[i];
            count = 0;
        }
        const size_t n2 = count / 2, stride = (size_t)gridDim.x * 512;
        const double2 *s2 = reinterpret_cast<const double2 *>(send);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n2; i += 8 * stride) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) if (i + u * stride < n2) v[u] = s2[i + u * stride];
#pragma unroll
            for (int u = 0; u < 8; u++) if (i + u * stride < n2) d2[i + u * stride] = v[u];
        }
        if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[count - 1] = send[count - 1];
    }
    __threadfence_system();                              // the slot's stores are out before the flag
    __syncthreads();
    if (lds_pad[(threadIdx.x * 7) & 511] < 0.0) me->broken = 2;              // (never true: keeps the LDS array alive)
    if (threadIdx.x == 0) {
        if (!ok) me->broken = 1;
        if (atomicAdd(&me->arrive_in, 1u) == gridDim.x - 1) {                // the last block publishes
            me->arrive_in = 0;
            __hip_atomic_store(&me->ready, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(512) void k_reduce(PeerTable P, int nranks, int rank, int root, double *__restrict__ recv, size_t count,
                                                unsigned long long k, unsigned long long ticks, int vec)
{
    __shared__ double lds_pad[512];                      // 4 KB: an RCCL-like LDS footprint (keeps the block off CUs whose LDS is full)
    __shared__ int ok;
    DevHeader *me = reinterpret_cast<DevHeader *>(P.buf[rank]);
    if (threadIdx.x == 0) ok = wait_flags(P, nranks, 0, k + 1, ticks) ? 1 : 0;
    lds_pad[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    if (ok) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");    // the peers' slots were written by other processes' kernels
        if (!vec) {
            for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < count; i += (size_t)gridDim.x * 512) {
                double sl = dslot(P, root >= 0 ? root : 0, k)[i];
                if (root < 0) for (int r = 1; r < nranks; r++) sl += dslot(P, r, k)[i];
                recv[i] = sl;
            }
            count = 0;
        }
        const size_t n2 = count / 2, stride = (size_t)gridDim.x * 512;
        double2 *r2 = reinterpret_cast<double2 *>(recv);
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n2; i += 4 * stride) {
            double2 acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i + u * stride < n2) acc[u] = reinterpret_cast<const double2 *>(dslot(P, root >= 0 ? root : 0, k))[i + u * stride];
            if (root < 0)
                for (int r = 1; r < nranks; r++) {                                  // rank order: every rank forms the same bits
                    const double2 *p2 = reinterpret_cast<const double2 *>(dslot(P, r, k));
#pragma unroll
                    for (int u = 0; u < 4; u++) if (i + u * stride < n2) { const double2 v = p2[i + u * stride]; acc[u].x += v.x; acc[u].y += v.y; }
                }
#pragma unroll
            for (int u = 0; u < 4; u++) if (i + u * stride < n2) r2[i + u * stride] = acc[u];
        }
        if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            double sl = dslot(P, root >= 0 ? root : 0, k)[count - 1];
            if (root < 0) for (int r = 1; r < nranks; r++) sl += dslot(P, r, k)[count - 1];
            recv[count - 1] = sl;
        }
    }
    if (lds_pad[(threadIdx.x * 7) & 511] < 0.0) recv[0] = 0.0;                      // (never true: keeps the LDS array alive)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!ok) me->broken = 1;
        if (atomicAdd(&me->arrive_red, 1u) == gridDim.x - 1) {
            me->arrive_red = 0;
            __hip_atomic_store(&me->done, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct Comm {
    bool kernel_mode = false;
    char *devbuf = nullptr;                              // this rank's exchange buffer (device memory, IPC-shared)
    PeerTable peers{};
    int red_blocks = 24;
    int nranks = 0, rank = 0;
    std::string path;
    char *base = nullptr;
    size_t bytes = 0;
    Header *h = nullptr;
    bool pinned = false;
    double *result = nullptr;                            // private pinned staging buffer of the reduced data
    uint64_t ops = 0;                                    // collectives enqueued so far (host thread)
    std::atomic<int> broken{0};
    double timeout_s = 60.0;
};

struct Op {
    Comm *c;
    uint64_t k;
    size_t count;
    int root;                                            // < 0: all-reduce (sum); >= 0: broadcast from root
};

char *slot(Comm *c, int r) { return c->base + kHeaderBytes + (size_t)r * kSlotBytes; }

bool wait_all(Comm *c, std::atomic<uint64_t> *arr, uint64_t want)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; spins++) {
        bool ok = true;
        for (int r = 0; r < c->nranks; r++) ok = ok && arr[r].load(std::memory_order_acquire) >= want;
        if (ok) return true;
        if (c->broken.load()) return false;
        if ((spins & 0x3FF) == 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
                c->broken.store(1);
                fprintf(stderr, "[fake_rccl] rank %d gave up waiting for its peers after %.0f s\n", c->rank, c->timeout_s);
                return false;
            }
            std::this_thread::yield();
        }
    }
}

void host_reduce(void *arg)
{
    Op *op = static_cast<Op *>(arg);
    Comm *c = op->c;
    c->h->ready[c->rank].store(op->k + 1, std::memory_order_release);
    if (wait_all(c, c->h->ready, op->k + 1)) {
        if (op->root >= 0) {
            std::memcpy(c->result, slot(c, op->root), op->count * sizeof(double));
        } else {
            const double *s0 = reinterpret_cast<const double *>(slot(c, 0));
            for (size_t i = 0; i < op->count; i++) c->result[i] = s0[i];
            for (int r = 1; r < c->nranks; r++) {                       // rank order: every rank forms the same bits
                const double *s = reinterpret_cast<const double *>(slot(c, r));
                for (size_t i = 0; i < op->count; i++) c->result[i] += s[i];
            }
        }
        c->h->read[c->rank].store(op->k + 1, std::memory_order_release);
        (void)wait_all(c, c->h->read, op->k + 1);
    }
    delete op;
}

ncclResult_t collective(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || dt != ncclDouble) return ncclInvalidArgument;
    if (count * sizeof(double) > kSlotBytes) return ncclInvalidArgument;
    if (c->broken.load()) return ncclSystemError;
    if (c->kernel_mode) {
        if (count * sizeof(double) > kDevSlotBytes) return ncclInvalidArgument;
        const unsigned long long k = c->ops++;
        const unsigned long long ticks = (unsigned long long)(c->timeout_s * 1e8);          // wall_clock64: 100 MHz
        const int vec = (((uintptr_t)send | (uintptr_t)recv) & 15) == 0 ? 1 : 0;
        const int nb_in = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->red_blocks, (count / 2 + 511) / 512));
        hipLaunchKernelGGL(k_copy_in, dim3(nb_in), dim3(512), 0, stream, c->peers, c->nranks, c->rank, static_cast<const double *>(send), count, k, ticks, vec);
        hipLaunchKernelGGL(k_reduce, dim3(c->red_blocks), dim3(512), 0, stream, c->peers, c->nranks, c->rank, root, static_cast<double *>(recv), count, k, ticks, vec);
        return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    }
    Op *op = new Op{c, c->ops++, count, root};
    if (hipMemcpyAsync(slot(c, c->rank), send, count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, host_reduce, op) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, c->result, count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{0};
    std::memset(id, 0, sizeof *id);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "fake_rccl_%d_%u_%llx", (int)getpid(), counter++, (unsigned long long)now);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    Comm *c = new Comm();
    c->nranks = nranks; c->rank = rank;
    if (const char *s = getenv("FAKE_RCCL_TIMEOUT_S")) { const double v = atof(s); if (v > 0) c->timeout_s = v; }
    c->path = std::string("/dev/shm/") + id.internal;
    c->bytes = kHeaderBytes + (size_t)nranks * kSlotBytes;
    const int fd = open(c->path.c_str(), O_CREAT | O_RDWR, 0600);      // whoever comes first creates it (zero-filled)
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    c->base = static_cast<char *>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
    c->h = reinterpret_cast<Header *>(c->base);
    // pinning this rank's slot makes the device -> host copy a true asynchronous, stream-ordered DMA
    c->pinned = hipHostRegister(slot(c, rank), kSlotBytes, hipHostRegisterDefault) == hipSuccess;
    if (!c->pinned) (void)hipGetLastError();
    if (hipHostMalloc(reinterpret_cast<void **>(&c->result), kSlotBytes, hipHostMallocDefault) != hipSuccess) {
        munmap(c->base, c->bytes); delete c; return ncclUnhandledCudaError;
    }
    c->h->arrived.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();                  // like the real call: returns once every rank is there
    while (c->h->arrived.load() < (uint32_t)nranks) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            fprintf(stderr, "[fake_rccl] rank %d: only %u of %d ranks arrived\n", rank, c->h->arrived.load(), nranks);
            (void)hipHostFree(c->result); munmap(c->base, c->bytes); delete c;
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (rank == 0) unlink(c->path.c_str());                            // every rank holds its mapping: the name can go
    const char *km = getenv("FAKE_RCCL_KERNEL");
    if (km && km[0] == '1') {
        // the exchange buffers: allocate, publish the IPC handle, open every peer's
        if (const char *b = getenv("FAKE_RCCL_BLOCKS")) { const int v = atoi(b); if (v >= 1 && v <= 256) c->red_blocks = v; }
        const size_t devbytes = kDevHeaderBytes + 2 * kDevSlotBytes;
        bool ok = hipMalloc(reinterpret_cast<void **>(&c->devbuf), devbytes) == hipSuccess && hipMemset(c->devbuf, 0, devbytes) == hipSuccess &&
                  hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(&c->h->ipc[rank], c->devbuf) == hipSuccess;
        if (ok) c->h->ipc_ready[rank].store(1, std::memory_order_release);
        const auto t1 = std::chrono::steady_clock::now();
        for (int r = 0; ok && r < nranks; r++) {
            while (!c->h->ipc_ready[r].load(std::memory_order_acquire)) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() > c->timeout_s) { ok = false; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (!ok) break;
            if (r == rank) c->peers.buf[r] = c->devbuf;
            else ok = hipIpcOpenMemHandle(reinterpret_cast<void **>(&c->peers.buf[r]), c->h->ipc[r], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
        }
        if (!ok) {
            fprintf(stderr, "[fake_rccl] rank %d: the IPC exchange buffers could not be set up (%s)\n", rank, hipGetErrorString(hipGetLastError()));
            return ncclUnhandledCudaError;
        }
        c->h->ipc_opened.fetch_add(1);
        while (c->h->ipc_opened.load() < (uint32_t)nranks) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        c->kernel_mode = true;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    c->broken.store(1);                                                // a host function still waiting leaves at once
    c->h->left.fetch_add(1);
    if (c->kernel_mode) {
        (void)hipDeviceSynchronize();
        for (int r = 0; r < c->nranks; r++) if (r != c->rank && c->peers.buf[r]) (void)hipIpcCloseMemHandle(c->peers.buf[r]);
        // (the owner's buffer stays allocated until every peer has left: a peer's kernel may still poll its flags)
        const auto t0 = std::chrono::steady_clock::now();
        while (c->h->left.load() < (uint32_t)c->nranks && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        (void)hipFree(c->devbuf);
    }
    if (c->pinned) (void)hipHostUnregister(slot(c, c->rank));
    (void)hipHostFree(c->result);
    munmap(c->base, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (op != ncclSum) return ncclInvalidArgument;
    return collective(send, recv, count, dt, -1, comm, stream);
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    return collective(send, recv, count, dt, root, comm, stream);
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
        case ncclSystemError: return "fake_rccl: system error (shared memory, or a peer never arrived)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"
