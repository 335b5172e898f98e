// engine.hip -- the device-resident VB-NMF engine behind the C ABI of include/vbnmf.h.
//
// One step (reference src/vbnmf_update.cpp:33-90) is four launches on one HIP stream:
//   k_update(W)  k_update(H)  k_sweep(both sides)  k_final
// The sweep at the end of step t produces the sufficient statistics that step t+1 starts
// from AND the data term of step t's evidence (see kernels.h), so X is streamed once per
// step and side.  k_final leaves lkh and the hyper statistics in pinned host memory and
// raises a sequence flag the host polls (no memcpy, no stream synchronise on the hot path).
// A cell-partitioned engine adds k_pack + k_tail, which fill the reduce buffer
// [swsum | rowSum(eh) | scalars] that the caller all-reduces between step_local and step_finish.
// vbnmf_engine_run drives the whole loop of vb_iterate from the device (hyper_update, stopping rule, per-step history;
// steps queued ahead of the GPU): the control step is folded into the next step's gene-side update (kernels.h:
// ControlFold; three launches per step), partitioned engines run it as the one-block kernel k_control behind each sweep.
// The same engine also runs the maximum-likelihood NMF step of factorize() (mlnmf.h: two single-side sweeps
// per step) and the sparse products of the svd2 initialiser's truncated SVD (k_spmm).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <queue>
#include <string>
#include <thread>

#include "common.h"
#include "comm.h"
#include "kernels.h"
#include "mlnmf.h"
#include "init.h"

using namespace vbnmf;

#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return fail(VBNMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

namespace {

constexpr int kHostOut = 16;
void dev_free(void *p);             // returns a device buffer to the pool (below)          // doubles in the pinned result block of an engine

// The layout's arrays on the device.  Engines made from a cached layout (same matrix, same geometry, same device)
// share one of these; the per-engine part is DeviceSide::part.
struct DeviceArrays {
    int device = 0;
    uint32_t *packed = nullptr, *widx = nullptr, *task_major = nullptr, *inv_task = nullptr;
    double *wval = nullptr;
    int32_t *slice_width = nullptr, *seg_block = nullptr, *seg_ptr = nullptr, *wg_seg0 = nullptr, *inv_ptr = nullptr;
    int32_t *slice_fast = nullptr, *block_start = nullptr;
    int64_t *slice_off = nullptr;
    ~DeviceArrays()
    {
        int cur = -1;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        dev_free(packed); dev_free(widx); dev_free(wval); dev_free(task_major); dev_free(inv_task);
        dev_free(slice_width); dev_free(seg_block); dev_free(seg_ptr); dev_free(wg_seg0); dev_free(inv_ptr);
        dev_free(slice_off); dev_free(slice_fast); dev_free(block_start);
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};

struct DeviceSide {
    std::shared_ptr<DeviceArrays> arrays;      // owner of the pointers below (possibly shared with other engines)
    uint32_t *packed = nullptr, *widx = nullptr, *task_major = nullptr, *inv_task = nullptr;
    double *wval = nullptr;
    int32_t *slice_width = nullptr, *seg_block = nullptr, *seg_ptr = nullptr, *wg_seg0 = nullptr, *inv_ptr = nullptr;
    int32_t *slice_fast = nullptr, *block_start = nullptr;
    int64_t *slice_off = nullptr;
    double *part = nullptr;                    // this engine's per-task partial statistics
    int64_t n_major = 0, n_minor = 0, n_tasks = 0, n_slices = 0, n_slots = 0;
    int32_t block_width = 0, n_blocks = 0, n_wg = 0, row_slots = 0;
    bool wide = false;
};

// ---- device buffer pool.  An engine owns ~25 device buffers (state, partial rows up to 100 MB a side, reduce buffers);
// a rank sweep creates and destroys an engine per (run, rank) unit, and every hipFree synchronises the whole device --
// including the streams of other engines of the process (vb_factorize(concurrent = K)).  Freed buffers therefore go to a
// per-process free list and the next engine takes the smallest one that fits (within 1.5 x of the request); the list holds
// at most VBNMF_POOL_MB (default 4096) and gives the oldest buffers back to the driver beyond that.  Measured on the C4
// sweep (19 engines in a row): engine creation + destruction 0.5 s -> 0.1 s (profiles/r04_c4_sweep.json).
struct PoolBuf { void *p; size_t bytes; int device; };
struct DevicePool {
    std::mutex mu;
    std::vector<PoolBuf> free_list;                  // oldest first
    std::vector<PoolBuf> live;                       // buffers handed out (size and device of every pointer)
    size_t held = 0;
    size_t cap = [] { const char *s = getenv("VBNMF_POOL_MB"); long v = s ? atol(s) : 4096; return (size_t)(v < 0 ? 0 : v) << 20; }();
};
DevicePool &device_pool() { static DevicePool *P = new DevicePool(); return *P; }      // (never destroyed: no HIP calls at exit)

int pool_alloc(void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    int device = 0;
    (void)hipGetDevice(&device);
    DevicePool &P = device_pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        int best = -1;
        for (int q = 0; q < (int)P.free_list.size(); q++) {
            const PoolBuf &b = P.free_list[q];
            if (b.device == device && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 + 4096 && (best < 0 || b.bytes < P.free_list[best].bytes)) best = q;
        }
        if (best >= 0) {
            PoolBuf b = P.free_list[best];
            P.free_list.erase(P.free_list.begin() + best);
            P.held -= b.bytes;
            P.live.push_back(b);
            *p = b.p;
            return VBNMF_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {                  // give the pooled buffers back and try once more
        (void)hipGetLastError();
        std::vector<PoolBuf> drop;
        { std::lock_guard<std::mutex> g(P.mu); drop.swap(P.free_list); P.held = 0; }
        for (const PoolBuf &b : drop) (void)hipFree(b.p);
        e = hipMalloc(p, bytes);
    }
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(VBNMF_ERR_OOM, "out of device memory (%zu bytes)", bytes); }
    if (e != hipSuccess) return fail(VBNMF_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> g(P.mu);
    P.live.push_back({*p, bytes, device});
    return VBNMF_OK;
}

// The caller guarantees that no queued work still uses the buffer (engines synchronise their streams before they free).
void dev_free(void *p)
{
    if (!p) return;
    DevicePool &P = device_pool();
    std::vector<PoolBuf> drop;
    {
        std::lock_guard<std::mutex> g(P.mu);
        PoolBuf b{p, 0, 0};
        bool known = false;
        for (size_t q = 0; q < P.live.size(); q++)
            if (P.live[q].p == p) { b = P.live[q]; P.live.erase(P.live.begin() + q); known = true; break; }
        if (!known || P.cap == 0 || b.bytes > P.cap) { drop.push_back(b); }
        else {
            P.free_list.push_back(b);
            P.held += b.bytes;
            while (P.held > P.cap && !P.free_list.empty()) {
                drop.push_back(P.free_list.front());
                P.held -= P.free_list.front().bytes;
                P.free_list.erase(P.free_list.begin());
            }
        }
    }
    for (const PoolBuf &b : drop) (void)hipFree(b.p);
}

template <typename T>
int dev_alloc(T **p, size_t count)
{
    if (count == 0) count = 1;
    return pool_alloc((void **)p, count * sizeof(T));
}

template <typename T>
int dev_upload(T **p, const ExtVec<T> &v)
{
    if (int rc = dev_alloc(p, v.size())) return rc;
    if (!v.empty()) HIPCHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return VBNMF_OK;
}

template <typename T, typename A>
int dev_upload(T **p, const std::vector<T, A> &v)
{
    if (int rc = dev_alloc(p, v.size())) return rc;
    if (!v.empty()) HIPCHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return VBNMF_OK;
}

}  // namespace

namespace vbnmf {
// vbnmf_set_engine_grid: the grids of the engines THIS host thread creates next (0: the defaults)
static thread_local int tl_grid_nwg = 0, tl_grid_ub = 0;
// vbnmf_set_engine_padding: the padded rank (row width) of the engines THIS host thread creates next (0: the rank's own)
static thread_local int tl_pad_R = 0;

int sweep_workgroups(int device, bool partitioned, int &n_wg)
{
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));
    n_wg = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;      // one persistent workgroup per CU
    const char *forced = getenv("VBNMF_NWG");
    if (forced) { int v = atoi(forced); if (v > 0) n_wg = v; }
    if (partitioned && !forced) {
        // A partition's sweep leaves CUs free: its workgroups own all 160 KB of a CU's LDS, so the all-reduce kernels that
        // should run BESIDE the cell-side sweep could not start on a chip filled by it.  How many: ONE PER SHADER ENGINE
        // (8 XCDs x 4 engines = 32).  Measured in round 5 with the collective carried by kernels of RCCL's launch shape
        // (tests/fake_rccl, FAKE_RCCL_KERNEL=1: 24 or 64 blocks x 512 threads, 4 KB of LDS; profiles/r05_c5_overlap.txt, one C5
        // partition): with 8 or 16 CUs free the collective's first kernel still ENDS WITH the sweep (155-166 us instead of
        // 25 alone) -- a queue's workgroups are dealt to the shader engines in turn, and the first one dealt to an engine
        // whose CUs are all full stalls the whole queue behind it -- with 32 free it runs inside the sweep (37 + 40 us,
        // done 170 us before the sweep ends).  Price: the two sweeps on 224 instead of 248 workgroups, +21 us per step
        // (0.427 against 0.411 ms with no reserve and this one-rank collective trailing the sweep): the reserve pays as soon
        // as the real all-reduce over xGMI takes more than ~25 us alone (4.8 MB: >= 55 us by SURVEY.md section 5's ring estimate).
        // VBNMF_COMM_CUS overrides (0: no reserve).
        int spare = 32;
        if (const char *sv = getenv("VBNMF_COMM_CUS")) spare = atoi(sv);
        if (spare >= 0 && n_wg - spare >= 8) n_wg -= spare;
    }
    return VBNMF_OK;
}
}  // namespace vbnmf

struct vbnmf_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, m = 0, m_global = 0, nnz = 0;
    int64_t col_begin = 0;            // first cell of this partition (0 for an unpartitioned engine)
    // The layout's internal renumbering of this engine's cells (csrc/order.cpp): the cell-indexed arrays (lh, llh, eh, dh,
    // the labels) hold position p = original local column cell_perm[p]; empty = as stored.  Every boundary that takes or
    // returns cell-indexed data translates through it; d_perm is its device copy (random_state's per-element counters).
    std::vector<int32_t> cell_perm;
    int32_t *d_perm = nullptr;
    int32_t *d_ids[2] = {nullptr, nullptr};   // arg-max labels of the cells: current / previous (cluster_changes)
    int ids_cur = 0;
    bool ids_valid = false;
    unsigned long long *d_table = nullptr;    // [(r+1)^2 + 1] contingency table of two labelings, then the pair count
    double *svd_ws = nullptr;         // truncated SVD work space: Gram block partials, two k x k matrices, values
    int32_t *svd_status = nullptr;
    int r = 0, R = 0, NT = 0, n_wg = 0;
    bool wide = false, partitioned = false;
    double lgx = 0.0;
    double xlx = 0.0;                 // sum over stored entries of -x log x + x (ML-NMF likelihood constant)
    DeviceSide A, B;                  // A: lanes own genes ; B: lanes own cells
    double *lw = nullptr, *llw = nullptr, *ew = nullptr, *dw = nullptr;
    double *lh = nullptr, *llh = nullptr, *eh = nullptr, *dh = nullptr;
    double *epart = nullptr;          // [2 * n_wg] evidence partials: gene side, then cell side
    double *bpW = nullptr, *bpH = nullptr;   // [ub][R+2] block partials of the two updates (allocated for kUpdateBlocks rows)
    // Both posterior updates in one launch (kernels.h: k_update2; unpartitioned engines): the gene side of the sweep leaves
    // per-slice column sums of sw (csl) and their per-workgroup sums (csum); both tables of block partials alternate.
    double *csl = nullptr, *csum = nullptr;  // [A.n_slices][R], [n_wg][R]
    bool pair = false;
    bool stream_nt = false;                  // the sweeps read the entry stream non-temporally (kernels.h: ld_stream)
    uint4 *upd_tab = nullptr;                // k_update2's work table, one row per block (build_update_table)
    int32_t upd_stride4 = 0, upd_V = 0, upd_ids_off = 0;
    int ub = kUpdateBlocks;           // blocks of the update kernels (one per CU; VBNMF_UPDATE_BLOCKS for experiments)
    double *red = nullptr;            // [n*R | R+4]  (partitioned engines only use the first part)
    int64_t red_count = 0;
    bool epart_in_red = false;        // partitioned engines: the evidence partials live behind red[red_count) (kEvSlots + 1 doubles:
                                      // the second all-reduce of a device-driven step sends them as they are)
    double *red_g = nullptr;          // same shape: receive side of the all-reduce in a device-driven partitioned loop
    const double *red_in = nullptr;   // what the W update and the control kernel read: red (reduced in place by the
                                      // caller) or red_g (out of place, so that steps queued past the stop stay no-ops)
    vbnmf_comm *comm = nullptr;       // attached communicator (not owned)
    int comm_rank = 0;
    hipStream_t cstream = nullptr;    // the all-reduces of a partitioned loop are enqueued here, beside the cell-side sweep
    std::vector<hipEvent_t> ev_ring;  // events ordering the two streams, cycled (a step uses 3)
    size_t ev_next = 0;
    double *d_out = nullptr;          // [8]
    double *h_out = nullptr;          // pinned, device-visible [kHostOut]; [7] = sequence flag; [8..12] = hyper, lk0 of a device-driven loop
    double *h_out_dev = nullptr;      // device address of h_out
    double *h_stage = nullptr;        // pinned staging of set_state / get_state (index-major copies of the state arrays)
    size_t h_stage_count = 0;
    double *h_hist = nullptr;         // pinned, device-visible per-step history of the device-driven loops (grown on demand)
    double *h_hist_dev = nullptr;
    size_t h_hist_count = 0;
    double seq = 0.0;
    LogTabEntry *logtab = nullptr;    // [128] ln table of the sweep
    LoopCtl *ctl = nullptr;           // control block of the device-driven loop
    // Unpartitioned VB loop with the control step folded into the gene-side update (kernels.h: ControlFold): two control
    // blocks and a second table of gene-side block partials, alternating by step; bpW always names the table the latest
    // queued gene-side update writes (the one every later kernel reads), bpW_alt the other.
    LoopCtl *ctl2 = nullptr;
    double *bpW_alt = nullptr;
    double *bpH_alt = nullptr;           // the ML loop folds its control step into the H update: the same alternation for bpH
    const int32_t *stop_ptr = nullptr;   // the stop flag the kernels of the step being queued read (null: ctl->stop)
    int fold_step = 0;
    bool fold = false;
    bool run_active = false;
    unsigned long long *dbg = nullptr;   // diagnostic timestamps of the sweep (VBNMF_DEBUG_TIMES=1)
    size_t dbg_count = 0;
    size_t lds_bytes = 0;
    bool has_state = false, stats_ready = false, step_pending = false, prime_pending = false;
    bool poisoned = false;            // a wait on the device timed out: work may still be queued, nothing is waited for or freed
    bool ml_ready = false;            // lw / lh hold an ML-NMF state (w, h) and the cell-side statistics are current
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    bool ev_recorded = false, ev2_recorded = false;
    double sweep_ms = 0.0;
    int64_t sweep_launches = 0;
};

// doubles behind `red` / `red_g`: the reduce buffer proper (red_count, what the host-stepped exchange carries) and, for a
// partitioned engine, the evidence slots of the device-driven loop's second all-reduce (kEvSlots partials + sum lgamma(x+1))
inline int64_t red_alloc_count(const vbnmf_engine *e)
{
    return e->red_count + ((e->partitioned && 2 * (int64_t)e->n_wg <= kEvSlots) ? kEvSlots + 1 : 0);
}

namespace {

// Pinned staging buffer of an engine's state transfers (true asynchronous DMA instead of the runtime's bounce through its
// own staging for pageable memory).  Engines hand theirs on through a small per-process list when they are destroyed.
struct StagePool { std::mutex mu; std::vector<std::pair<double *, size_t>> free_list; };
StagePool &stage_pool() { static StagePool *P = new StagePool(); return *P; }

int ensure_stage(vbnmf_engine *e, size_t count)
{
    if (count <= e->h_stage_count) return VBNMF_OK;
    StagePool &P = stage_pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (e->h_stage) { P.free_list.emplace_back(e->h_stage, e->h_stage_count); e->h_stage = nullptr; e->h_stage_count = 0; }
        int best = -1;
        for (int q = 0; q < (int)P.free_list.size(); q++)
            if (P.free_list[q].second >= count && (best < 0 || P.free_list[q].second < P.free_list[best].second)) best = q;
        if (best >= 0) {
            e->h_stage = P.free_list[best].first; e->h_stage_count = P.free_list[best].second;
            P.free_list.erase(P.free_list.begin() + best);
            return VBNMF_OK;
        }
    }
    double *p = nullptr;
    hipError_t he = hipHostMalloc((void **)&p, count * sizeof(double), hipHostMallocDefault);
    if (he != hipSuccess) { (void)hipGetLastError(); return fail(VBNMF_ERR_OOM, "pinned staging buffer (%zu bytes): %s", count * sizeof(double), hipGetErrorString(he)); }
    e->h_stage = p; e->h_stage_count = count;
    return VBNMF_OK;
}

void release_stage(vbnmf_engine *e)
{
    if (!e->h_stage) return;
    StagePool &P = stage_pool();
    std::vector<std::pair<double *, size_t>> drop;
    {
        std::lock_guard<std::mutex> g(P.mu);
        P.free_list.emplace_back(e->h_stage, e->h_stage_count);
        size_t held = 0;
        for (auto &b : P.free_list) held += b.second * sizeof(double);
        while (P.free_list.size() > 8 || held > ((size_t)1 << 30)) {           // at most 8 buffers / 1 GiB kept
            held -= P.free_list.front().second * sizeof(double);
            drop.push_back(P.free_list.front());
            P.free_list.erase(P.free_list.begin());
        }
    }
    for (auto &b : drop) (void)hipHostFree(b.first);
    e->h_stage = nullptr; e->h_stage_count = 0;
}

void free_side(DeviceSide &S)
{
    dev_free(S.part);
    S = DeviceSide();                          // drops this engine's reference to the (shared) layout arrays
}

// Device copy of a layout's arrays; `X` != null: the layout is (possibly) cached on the matrix and so is its copy.
int upload_side(const Layout &L, int R, int device, const vbnmf_matrix *X, DeviceSide &S, bool with_part = true)
{
    S.n_major = L.n_major; S.n_minor = L.n_minor; S.n_tasks = L.n_tasks; S.n_slices = L.n_slices;
    S.n_slots = L.n_slots; S.block_width = L.block_width; S.n_blocks = L.n_blocks; S.n_wg = L.n_wg; S.wide = L.wide;
    S.row_slots = L.row_slots;
    if (S.row_slots < R / 2 / rank_shares(R) * rank_shares(R) || !(S.row_slots & 1))
        return fail(VBNMF_ERR_BAD_ARG, "layout row stride of %d slots cannot hold rows of padded rank %d", S.row_slots, R);
    std::shared_ptr<DeviceArrays> A;
    if (X) A = std::static_pointer_cast<DeviceArrays>(cached_device_copy(X, &L, device));
    if (!A) {
        A = std::make_shared<DeviceArrays>();
        A->device = device;
        if (L.wide) {
            if (int rc = dev_upload(&A->widx, L.wide_idx)) return rc;
            if (int rc = dev_upload(&A->wval, L.wide_val)) return rc;
        } else {
            if (int rc = dev_upload(&A->packed, L.packed)) return rc;
        }
        if (int rc = dev_upload(&A->task_major, L.task_major)) return rc;
        if (int rc = dev_upload(&A->slice_width, L.slice_width)) return rc;
        if (int rc = dev_upload(&A->slice_off, L.slice_off)) return rc;
        if (int rc = dev_upload(&A->slice_fast, L.slice_fast)) return rc;
        {
            std::vector<int32_t> bs(L.block_start.begin(), L.block_start.end());      // minors are < 2^31
            if (int rc = dev_upload(&A->block_start, bs)) return rc;
        }
        if (int rc = dev_upload(&A->seg_block, L.seg_block)) return rc;
        if (int rc = dev_upload(&A->seg_ptr, L.seg_ptr)) return rc;
        if (int rc = dev_upload(&A->wg_seg0, L.wg_seg0)) return rc;
        if (int rc = dev_upload(&A->inv_ptr, L.inv_ptr)) return rc;
        if (int rc = dev_upload(&A->inv_task, L.inv_task)) return rc;
        if (X) store_device_copy(X, &L, device, A);
    }
    S.arrays = A;
    S.packed = A->packed; S.widx = A->widx; S.wval = A->wval; S.task_major = A->task_major; S.inv_task = A->inv_task;
    S.slice_width = A->slice_width; S.seg_block = A->seg_block; S.seg_ptr = A->seg_ptr; S.wg_seg0 = A->wg_seg0;
    S.inv_ptr = A->inv_ptr; S.slice_off = A->slice_off; S.slice_fast = A->slice_fast; S.block_start = A->block_start;
    if (with_part) { if (int rc = dev_alloc(&S.part, (size_t)L.n_slices * kLanes * R)) return rc; }
    return VBNMF_OK;
}

SweepSide sweep_side_args(const vbnmf_engine *e, const DeviceSide &S, bool gene_side, double *epart)
{
    SweepSide P;
    P.packed = S.packed; P.widx = S.widx; P.wval = S.wval;
    P.task_major = S.task_major; P.slice_width = S.slice_width; P.slice_off = S.slice_off; P.slice_fast = S.slice_fast;
    P.seg_block = S.seg_block; P.wg_seg0 = S.wg_seg0; P.seg_ptr = S.seg_ptr;
    P.F = gene_side ? e->lw : e->lh;
    P.llF = gene_side ? e->llw : e->llh;
    P.G = gene_side ? e->lh : e->lw;
    P.part = S.part; P.epart = epart;
    P.csl = gene_side ? e->csl : nullptr; P.csum = gene_side ? e->csum : nullptr;
    P.n_minor = (int32_t)S.n_minor; P.block_start = S.block_start;
    P.row_slots = S.row_slots;
    {
        // The youngest third of the workgroup's waves pull slices from the SHORT end of a segment's list (kernels.h:
        // take_ticket_ends).  Measured per side, interleaved, on two boxes (k_sweep us, rank 10, 768 threads: neither side /
        // gene side only / cell side only / both): 171.5 / 170.0 / 175.5 / 172.5 and 168.7 / 166.7 / 171.9 / 169.9 -- the gene
        // side gains, the cell side (no logarithm, shorter slices) loses at this geometry; at the 512-thread ranks both
        // sides gain (neither / gene / both: rank 16 294.1 / 285.7 / 284.8, rank 20 378.9 / 369.0 / 365.3, rank 32 859 / 845 /
        // 838); below padded rank 8 it loses on both (rank 5: 121.1 / 127.3 / 129.3).  So: the gene side from padded rank
        // 8, the cell side from 16.  VBNMF_PULL_ENDS=k overrides the count on both sides, VBNMF_PULL_ENDS_A / _B per side.
        static const int pe = [] { const char *v = getenv("VBNMF_PULL_ENDS"); return v ? atoi(v) : -1; }();
        static const int pes[2] = {[] { const char *v = getenv("VBNMF_PULL_ENDS_A"); return v ? atoi(v) : -1; }(),
                                   [] { const char *v = getenv("VBNMF_PULL_ENDS_B"); return v ? atoi(v) : -1; }()};
        const int third = (e->NT / 64) / 3;
        int k = gene_side ? (e->R >= 8 ? third : 0) : (e->R >= 16 ? third : 0);
        if (pe >= 0) k = pe;
        if (pes[gene_side ? 0 : 1] >= 0) k = pes[gene_side ? 0 : 1];
        P.pull_ends = std::min(k, e->NT / 64);
    }
    P.logterm = gene_side ? 1 : 0;
    P.stream_nt = e->stream_nt ? 1 : 0;
    P.n_wg = S.n_wg;
    P.logtab = e->logtab;
    P.stop = e->run_active ? (e->stop_ptr ? e->stop_ptr : &e->ctl->stop) : nullptr;
    P.dbg = e->dbg ? e->dbg + (gene_side ? 0 : e->dbg_count / 2) : nullptr;
    return P;
}

// First use of a sweep kernel on a device: allow the 160 KB dynamic LDS image, and make sure the kernel has no
// static LDS in front of it -- lds_row() (kernels.h) addresses the staged block from LDS address 0.
int prepare_sweep_kernel(const void *fn)
{
    HIPCHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipFuncAttributes fa;
    HIPCHECK(hipFuncGetAttributes(&fa, fn));
    if (fa.sharedSizeBytes != 0)
        return fail(VBNMF_ERR_HIP, "sweep kernel carries %zu bytes of static LDS; its dynamic LDS would not start at 0", (size_t)fa.sharedSizeBytes);
    return VBNMF_OK;
}

// ---- dispatch over the padded rank (compile-time so factor rows live in registers) ----
template <int R, bool WIDE, int NT, int SP>
int launch_sweep_t(vbnmf_engine *e, const SweepSide &a, const SweepSide &b)
{
    static std::atomic<bool> attr_set[16];
    const void *fn = (const void *)k_sweep<R, WIDE, NT, SP>;
    if (e->device >= 16 || !attr_set[e->device].load(std::memory_order_acquire)) {
        if (int rc = prepare_sweep_kernel(fn)) return rc;
        if (e->device < 16) attr_set[e->device].store(true, std::memory_order_release);
    }
    const unsigned grid = (unsigned)e->n_wg;
    hipLaunchKernelGGL((k_sweep<R, WIDE, NT, SP>), dim3(grid), dim3(NT), e->lds_bytes, e->stream, a, b);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// RT: the padded rank; above 32 the sweep's lanes share it (SP lanes of R = RT / SP columns each, kernels.h)
template <int RT>
int launch_sweep_r(vbnmf_engine *e, const SweepSide &a, const SweepSide &b)
{
    constexpr int SP = rank_shares(RT), R = RT / SP, NT = sweep_threads(RT);
    return e->wide ? launch_sweep_t<R, true, NT, SP>(e, a, b) : launch_sweep_t<R, false, NT, SP>(e, a, b);
}

#define VBNMF_FOR_EACH_R_UP_TO_64(X) \
    X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
    X(40) X(48) X(56) X(64)
// every padded rank: up to 32 by 2 (one lane per task), 40..64 by 8 (two lanes), 80..128 by 16 (four lanes)
#ifdef VBNMF_DEV_FEW_RANKS              /* development builds only: a few ranks, for a compile check in a minute */
#define VBNMF_FOR_EACH_R(X) X(4) X(6) X(8) X(10) X(20) X(48) X(80)
#else
#define VBNMF_FOR_EACH_R(X) VBNMF_FOR_EACH_R_UP_TO_64(X) X(80) X(96) X(112) X(128)
#endif

int launch_sweep(vbnmf_engine *e)
{
    SweepSide a = sweep_side_args(e, e->A, true, e->epart);
    SweepSide b = sweep_side_args(e, e->B, false, e->epart + e->n_wg);
    if (e->timing) { HIPCHECK(hipEventRecord(e->ev0, e->stream)); }
    int rc = VBNMF_ERR_BAD_ARG;
    switch (e->R) {
#define X(RR) case RR: rc = launch_sweep_r<RR>(e, a, b); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    if (rc) return rc;
    if (e->timing) { HIPCHECK(hipEventRecord(e->ev1, e->stream)); e->ev_recorded = true; }
    return VBNMF_OK;
}

// ctl != nullptr: device-driven loop, the hyper-parameters are read from the control block on the device
int launch_update(vbnmf_engine *e, bool gene_side, double a, double b, double fudge, const LoopCtl *ctl = nullptr,
                  const ControlFold *foldp = nullptr)
{
    ControlFold fold{};
    if (foldp) fold = *foldp;
    const unsigned grid = fold.control_only ? 1 : e->ub;
    const double lga = (ctl || foldp) ? 0.0 : -std::lgamma(a) + a * std::log(a / b);     // reference :82 / :87
    const int side = gene_side ? 0 : 1;
    const bool dense = gene_side && e->partitioned;               // statistics already summed into `red`
    const DeviceSide &S = gene_side ? e->A : e->B;
    const double *redin = e->red_in ? e->red_in : e->red;
    const double *acc = dense ? redin : S.part;
    const int32_t *inv_ptr = dense ? nullptr : S.inv_ptr;
    const uint32_t *inv_task = dense ? nullptr : S.inv_task;
    const int64_t nmaj = gene_side ? e->n : e->m;
    const double *other = dense ? redin + (size_t)e->n * e->R : nullptr;
    const double *other_bp = dense ? nullptr : (gene_side ? e->bpH : e->bpW);
    const int other_nb = dense ? 0 : e->ub;
    double *l = gene_side ? e->lw : e->lh, *ll = gene_side ? e->llw : e->llh;
    double *ev = gene_side ? e->ew : e->eh, *d = gene_side ? e->dw : e->dh;
    double *bp = gene_side ? e->bpW : e->bpH;
    static const int stage_allowed = [] { const char *v = getenv("VBNMF_NO_STAGE_IDS"); return (v && v[0] == '1') ? 0 : 1; }();   // A/B switch
    // (staging the inverse index in LDS pays from a few hundred task ids per block on: on tiny matrices its two leading
    // round trips are all there is to the kernel -- 200 x 500 at rank 3: 30.7 -> 31.3 us per step with it)
    const int stage_ids = stage_allowed && !dense && S.n_tasks >= (int64_t)256 * grid ? 1 : 0;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_update<RR>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, acc, inv_ptr, inv_task, nmaj, e->r, other, other_bp, other_nb, a, b, lga, fudge, l, ll, ev, d, bp, ctl, side, fold, stage_ids); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// The work table of k_update2 (kernels.h): per block of the update, the visits of every thread row -- the block's genes
// and cells dealt to the rows longest-processing-time first -- and the task ids of those majors.  The gather is bound by the
// task rows it moves (a round of 16 scattered loads takes about as long as the whole posterior arithmetic behind it,
// in-kernel stamps), so a major costs its task count plus a few loads' worth of fixed work.  Every block's row has the same
// shape (stride, visits per thread row, offset of the ids), so the kernel loads it without knowing anything first.
// Returns false when a block's row does not fit the kernel's LDS copy: the engine then keeps the two-launch form.
bool build_update_table(const Layout &LA, const Layout &LB, int ub, int R, std::vector<uint32_t> &tab, int32_t &stride4, int32_t &V,
                        int32_t &ids_off)
{
    const int RB = kUpdateThreads / R;
    const int64_t nm[2] = {LA.n_major, LB.n_major};
    const Layout *Ls[2] = {&LA, &LB};
    const int64_t per[2] = {(nm[0] + ub - 1) / ub, (nm[1] + ub - 1) / ub};
    struct Item { int32_t cost; uint32_t code; int32_t cnt; };
    struct Plan { std::vector<std::vector<Item>> rows; int64_t ids = 0; int vmax = 0; };
    std::vector<Plan> plans(ub);
    // (a thread per 16 K majors: on a small matrix starting threads costs more than the table -- 1 ms of a 1.7 ms engine creation)
    const int table_threads = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (nm[0] + nm[1]) >> 14));
    parallel_for(ub, [&](int64_t b0, int64_t b1, int) {
        std::vector<Item> items;
        for (int64_t b = b0; b < b1; b++) {
            Plan &P = plans[b];
            items.clear();
            for (int sd = 0; sd < 2; sd++) {
                const int64_t m0 = std::min(nm[sd], b * per[sd]), m1 = std::min(nm[sd], m0 + per[sd]);
                for (int64_t M = m0; M < m1; M++) {
                    const int32_t cnt = Ls[sd]->inv_ptr[M + 1] - Ls[sd]->inv_ptr[M];
                    items.push_back({cnt + 4, ((uint32_t)sd << 31) | (uint32_t)(M - m0), cnt});
                    P.ids += cnt;
                }
            }
            std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.cost != y.cost ? x.cost > y.cost : x.code < y.code; });
            P.rows.assign(RB, {});
            // (load, row): the least loaded row takes the next item; ties go to the lowest row -- the plan is a function of the layouts alone
            std::priority_queue<std::pair<int64_t, int>, std::vector<std::pair<int64_t, int>>, std::greater<std::pair<int64_t, int>>> pq;
            for (int q = 0; q < RB; q++) pq.push({0, q});
            for (const Item &it : items) {
                auto top = pq.top(); pq.pop();
                P.rows[top.second].push_back(it);
                pq.push({top.first + it.cost, top.second});
            }
            for (int q = 0; q < RB; q++) {                  // a row visits its genes first, then its cells (k_update2's two loops)
                std::stable_sort(P.rows[q].begin(), P.rows[q].end(), [](const Item &x, const Item &y) { return (x.code >> 31) < (y.code >> 31); });
                P.vmax = std::max(P.vmax, (int)P.rows[q].size());
            }
        }
    }, table_threads);
    int64_t max_ids = 0;
    V = 1;
    for (const Plan &P : plans) { max_ids = std::max(max_ids, P.ids); V = std::max(V, (int32_t)P.vmax); }
    const int64_t vis_words = ((int64_t)RB * V * 3 + 3) & ~(int64_t)3;
    const int64_t stride = (vis_words + max_ids + 3) & ~(int64_t)3;
    if (stride > kUpdTabWords) return false;
    ids_off = (int32_t)vis_words;
    stride4 = (int32_t)(stride / 4);
    tab.assign((size_t)ub * stride, 0u);
    parallel_for(ub, [&](int64_t b0, int64_t b1, int) {
        std::vector<int32_t> first[2];                      // first id slot of each of the block's majors, by side
        for (int64_t b = b0; b < b1; b++) {
            uint32_t *row = tab.data() + (size_t)b * stride;
            int32_t q = 0;
            for (int sd = 0; sd < 2; sd++) {
                const int64_t m0 = std::min(nm[sd], b * per[sd]), m1 = std::min(nm[sd], m0 + per[sd]);
                first[sd].assign((size_t)(m1 - m0) + 1, 0);
                for (int64_t M = m0; M < m1; M++) {
                    first[sd][M - m0] = q;
                    for (int32_t u = Ls[sd]->inv_ptr[M]; u < Ls[sd]->inv_ptr[M + 1]; u++) row[ids_off + q++] = Ls[sd]->inv_task[u];
                }
                first[sd][m1 - m0] = q;
            }
            const Plan &P = plans[b];
            for (int rr = 0; rr < RB; rr++) {
                uint32_t *vis = row + (size_t)rr * V * 3;
                int v = 0;
                for (const Item &it : P.rows[rr]) {
                    const int sd = (int)(it.code >> 31);
                    const uint32_t loc = it.code & 0x7FFFFFFFu;
                    vis[3 * v] = it.code; vis[3 * v + 1] = (uint32_t)first[sd][loc]; vis[3 * v + 2] = (uint32_t)(first[sd][loc] + it.cnt);
                    v++;
                }
                for (; v < V; v++) { vis[3 * v] = 0xFFFFFFFFu; vis[3 * v + 1] = 0; vis[3 * v + 2] = 0; }
            }
        }
    }, table_threads);
    return true;
}

// Both posterior updates in one launch (kernels.h: k_update2).  Both tables of block partials alternate: the launch reads
// the previous ones and writes the others; e->bpW / e->bpH always name the latest.
int launch_update2(vbnmf_engine *e, double aw, double bw, double ah, double bh, double fudge, const LoopCtl *ctl = nullptr,
                   const ControlFold *foldp = nullptr)
{
    ControlFold fold{};
    if (foldp) fold = *foldp;
    const unsigned grid = (unsigned)e->ub;
    UpdSide W{}, H{};
    W.part = e->A.part; W.nmaj = e->n;
    W.l = e->lw; W.ll = e->llw; W.e = e->ew; W.d = e->dw;
    H.part = e->B.part; H.nmaj = e->m;
    H.l = e->lh; H.ll = e->llh; H.e = e->eh; H.d = e->dh;
    UpdTable T{e->upd_tab, e->upd_stride4, e->upd_V, e->upd_ids_off};
    W.bp_prev = e->bpW; H.bp_prev = e->bpH;
    std::swap(e->bpW, e->bpW_alt); std::swap(e->bpH, e->bpH_alt);
    W.bp = e->bpW; H.bp = e->bpH;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_update2<RR>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, W, H, T, e->r, e->ub, (const double *)e->csum, e->n_wg, aw, bw, ah, bh, fudge, ctl, fold); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_prime(vbnmf_engine *e, bool gene_side)
{
    const int64_t nmaj = gene_side ? e->n : e->m;
    const double *l = gene_side ? e->lw : e->lh;
    double *ll = gene_side ? e->llw : e->llh;
    const double *ev = gene_side ? nullptr : e->eh;
    double *bp = gene_side ? e->bpW : e->bpH;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_prime<RR>), dim3(e->ub), dim3(kUpdateThreads), 0, e->stream, nmaj, e->r, l, ll, ev, bp); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_final(vbnmf_engine *e)
{
    const double *tail = e->partitioned ? e->red + (size_t)e->n * e->R : nullptr;
    const int64_t nep = 2 * (int64_t)e->n_wg;
    e->seq += 1.0;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_final<RR>), dim3(1), dim3(1024), 0, e->stream, e->bpW, e->ub, tail, e->bpH, e->ub, e->epart, nep, e->lgx, e->r, (double)e->n, (double)e->m_global, e->seq, e->d_out, e->h_out_dev); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// partitioned engines: sweep output -> reduce buffer = [swsum | rowSum(eh) | sum H-terms | sum log lh | data term | lgx]
int launch_pack(vbnmf_engine *e)
{
    const int64_t cnt = e->n * e->R;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, e->A.part, e->A.inv_ptr, e->A.inv_task, e->n, e->R, e->red);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), 0, e->stream, e->bpH, e->ub, e->R, e->epart,
                       2 * (int64_t)e->n_wg, e->lgx, e->red + cnt);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// ---- ML-NMF (mlnmf.h): single-side sweeps and the multiplicative updates ----
template <int R, bool WIDE, bool LOGTERM, int NT, int SP, bool VB = false>
int launch_sweep1_t(vbnmf_engine *e, const SweepSide &a)
{
    static std::atomic<bool> attr_set[16];
    const void *fn = (const void *)k_sweep1<R, WIDE, LOGTERM, NT, VB, SP>;
    if (e->device >= 16 || !attr_set[e->device].load(std::memory_order_acquire)) {
        if (int rc = prepare_sweep_kernel(fn)) return rc;
        if (e->device < 16) attr_set[e->device].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_sweep1<R, WIDE, LOGTERM, NT, VB, SP>), dim3((unsigned)e->n_wg), dim3(NT), e->lds_bytes, e->stream, a);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// One side of the VB sweep alone (cell-partitioned device loop): gene side with the log term, cell side without.
template <int RT>
int launch_vb_side_r(vbnmf_engine *e, const SweepSide &a, bool gene_side)
{
    constexpr int SP = rank_shares(RT), R = RT / SP, NT = sweep_threads(RT);
    if (e->wide) return gene_side ? launch_sweep1_t<R, true, true, NT, SP, true>(e, a) : launch_sweep1_t<R, true, false, NT, SP, true>(e, a);
    return gene_side ? launch_sweep1_t<R, false, true, NT, SP, true>(e, a) : launch_sweep1_t<R, false, false, NT, SP, true>(e, a);
}

template <int RT>
int launch_sweep1_r(vbnmf_engine *e, const SweepSide &a, bool logterm)
{
    constexpr int SP = rank_shares(RT), R = RT / SP, NT = sweep_threads(RT);
    if (e->wide) return logterm ? launch_sweep1_t<R, true, true, NT, SP>(e, a) : launch_sweep1_t<R, true, false, NT, SP>(e, a);
    return logterm ? launch_sweep1_t<R, false, true, NT, SP>(e, a) : launch_sweep1_t<R, false, false, NT, SP>(e, a);
}

// gene side: lanes own genes (F = w, G = h), statistics for the W update; cell side: F = h, G = w, statistics
// for the H update and sum x log(wh) in the cell half of epart.
int launch_sweep1(vbnmf_engine *e, bool gene_side)
{
    SweepSide a = sweep_side_args(e, gene_side ? e->A : e->B, gene_side, gene_side ? e->epart : e->epart + e->n_wg);
    a.logterm = gene_side ? 0 : 1;
    hipEvent_t t0 = gene_side ? e->ev0 : e->ev2, t1 = gene_side ? e->ev1 : e->ev3;
    if (e->timing) { HIPCHECK(hipEventRecord(t0, e->stream)); }
    int rc = VBNMF_ERR_BAD_ARG;
    switch (e->R) {
#define X(RR) case RR: rc = launch_sweep1_r<RR>(e, a, !gene_side); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    if (rc) return rc;
    if (e->timing) { HIPCHECK(hipEventRecord(t1, e->stream)); (gene_side ? e->ev_recorded : e->ev2_recorded) = true; }
    return VBNMF_OK;
}

int launch_vb_side(vbnmf_engine *e, bool gene_side)
{
    SweepSide a = sweep_side_args(e, gene_side ? e->A : e->B, gene_side, gene_side ? e->epart : e->epart + e->n_wg);
    int rc = VBNMF_ERR_BAD_ARG;
    switch (e->R) {
#define X(RR) case RR: rc = launch_vb_side_r<RR>(e, a, gene_side); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    return rc;
}

int launch_ml_update(vbnmf_engine *e, bool gene_side, int prior, double ga, double gb, double eps, const MlFold *foldp = nullptr)
{
    const DeviceSide &S = gene_side ? e->A : e->B;
    const int64_t nmaj = gene_side ? e->n : e->m;
    const double *other_bp = gene_side ? e->bpH : e->bpW;
    double *f = gene_side ? e->lw : e->lh;
    double *bp = gene_side ? e->bpW : e->bpH;
    const int32_t *stop = e->run_active ? (e->stop_ptr ? e->stop_ptr : &e->ctl->stop) : nullptr;
    MlFold fold{};
    if (foldp) fold = *foldp;
    const unsigned grid = fold.control_only ? 1 : (unsigned)e->ub;
    static const int stage_allowed = [] { const char *v = getenv("VBNMF_NO_STAGE_IDS"); return (v && v[0] == '1') ? 0 : 1; }();
    const int stage_ids = stage_allowed && S.n_tasks >= (int64_t)256 * grid ? 1 : 0;      // (as launch_update)
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_ml_update<RR>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, S.part, S.inv_ptr, S.inv_task, nmaj, e->r, other_bp, e->ub, prior, ga, gb, eps, f, bp, stop, fold, stage_ids); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_ml_final(vbnmf_engine *e)
{
    e->seq += 1.0;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_ml_final<RR>), dim3(1), dim3(1024), 0, e->stream, e->bpW, e->bpH, e->ub, e->epart + e->n_wg, (int64_t)e->n_wg, e->xlx, e->r, (double)e->n, (double)e->m, e->seq, e->d_out, e->h_out_dev); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// ---- sparse products on the tiled layout (k_spmm) ----
template <int R, bool WIDE, int NT, int SP>
int launch_spmm_t(vbnmf_engine *e, const SweepSide &a)
{
    static std::atomic<bool> attr_set[16];
    const void *fn = (const void *)k_spmm<R, WIDE, NT, SP>;
    if (e->device >= 16 || !attr_set[e->device].load(std::memory_order_acquire)) {
        if (int rc = prepare_sweep_kernel(fn)) return rc;
        if (e->device < 16) attr_set[e->device].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_spmm<R, WIDE, NT, SP>), dim3((unsigned)e->n_wg), dim3(NT), e->lds_bytes, e->stream, a);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

template <int RT>
int launch_spmm_r(vbnmf_engine *e, const SweepSide &a)
{
    constexpr int SP = rank_shares(RT), R = RT / SP, NT = sweep_threads(RT);
    return e->wide ? launch_spmm_t<R, true, NT, SP>(e, a) : launch_spmm_t<R, false, NT, SP>(e, a);
}

// Wall-clock bound of the host's waits on the device (seconds): VBNMF_WAIT_TIMEOUT_S, default 300.  A wait that
// exceeds it returns VBNMF_ERR_HIP instead of spinning for ever -- e.g. a cell-partitioned run whose peer died or never
// enqueued its collective leaves this process's stream on hipErrorNotReady for good (failure modes).
double wait_timeout_s()
{
    if (const char *s = getenv("VBNMF_WAIT_TIMEOUT_S")) {
        const double v = atof(s);
        if (v > 0.0) return v;
    }
    return 300.0;
}

// Wait for k_final's sequence flag in pinned memory; falls back to the stream if it takes long.
int wait_result(vbnmf_engine *e)
{
    volatile double *flag = e->h_out + 7;
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = wait_timeout_s();
    for (long spins = 0; *flag != e->seq; spins++) {
        if (spins > 0 && (spins & 0xFFFF) == 0) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > limit) {
                e->poisoned = true;
                return fail(VBNMF_ERR_HIP, "timed out after %.1f s (VBNMF_WAIT_TIMEOUT_S) waiting for step %.0f; the last completed step is %.0f",
                            waited, e->seq, (double)*flag);
            }
            hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) {
                if (*flag == e->seq) break;
                HIPCHECK(hipStreamSynchronize(e->stream));
                if (*flag != e->seq) return fail(VBNMF_ERR_HIP, "the step finished but its result flag was never raised");
                break;
            }
            if (q != hipErrorNotReady) return fail(VBNMF_ERR_HIP, "the step failed on the device: %s", hipGetErrorString(q));
        }
    }
    return VBNMF_OK;
}

// hipStreamSynchronize with the bound of wait_timeout_s(): for streams that may sit behind a collective whose peer is
// gone (partitioned engines).  On a timeout the engine is poisoned like the bounded waits of the step paths.
int bounded_stream_sync(vbnmf_engine *e, hipStream_t stream, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = wait_timeout_s();
    for (long spins = 0;; spins++) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return VBNMF_OK;
        if (q != hipErrorNotReady) return fail(VBNMF_ERR_HIP, "%s failed on the device: %s", what, hipGetErrorString(q));
        if ((spins & 0xFF) == 0xFF) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > limit) {
                e->poisoned = true;
                return fail(VBNMF_ERR_HIP, "timed out after %.1f s (VBNMF_WAIT_TIMEOUT_S) waiting for %s", waited, what);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
}

int harvest_timing(vbnmf_engine *e)
{
    if (e->timing && e->ev_recorded) {
        float ms = 0.f;
        HIPCHECK(hipEventSynchronize(e->ev1));
        HIPCHECK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->sweep_ms += ms;
        e->sweep_launches++;
        e->ev_recorded = false;
    }
    if (e->timing && e->ev2_recorded) {
        float ms = 0.f;
        HIPCHECK(hipEventSynchronize(e->ev3));
        HIPCHECK(hipEventElapsedTime(&ms, e->ev2, e->ev3));
        e->sweep_ms += ms;
        e->sweep_launches++;
        e->ev2_recorded = false;
    }
    return VBNMF_OK;
}

int use_device(const vbnmf_engine *e)
{
    // A bounded wait gave up earlier: work may still sit on the engine's streams behind a collective whose peer is gone,
    // so any call that synchronises (get_state's memcpy, set_state, another run) would block without bound -- the very
    // hang the timeout exists to prevent.  Every entry point passes through here: fail fast, only destroy is left.
    if (e->poisoned)
        return fail(VBNMF_ERR_STATE, "the engine timed out earlier (VBNMF_WAIT_TIMEOUT_S) and may still have work queued; destroy it");
    HIPCHECK(hipSetDevice(e->device));
    return VBNMF_OK;
}

int check_device(int device)
{
    int cnt = 0;
    hipError_t err = hipGetDeviceCount(&cnt);
    if (err != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        return fail(VBNMF_ERR_NO_DEVICE, "no HIP device is available (%s); this library has no CPU fallback",
                    err != hipSuccess ? hipGetErrorString(err) : "device count is 0");
    }
    if (device < 0 || device >= cnt) return fail(VBNMF_ERR_BAD_ARG, "device %d is outside [0, %d)", device, cnt);
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VBNMF_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    return VBNMF_OK;
}

// Statistics of a freshly loaded state (lw, lh, eh on the device): rowSum(eh), then the sweep (its evidence partials
// are not used); a partitioned engine leaves them in the reduce buffer for the exchange (state_finish follows).
int prime_state(vbnmf_engine *e)
{
    e->has_state = true;
    e->ids_valid = false;
    if (int rc = launch_prime(e, true)) return rc;
    if (int rc = launch_prime(e, false)) return rc;
    if (int rc = launch_sweep(e)) return rc;
    if (e->partitioned) { if (int rc = launch_pack(e)) return rc; }
    e->prime_pending = true;
    if (!e->partitioned) return vbnmf_engine_state_finish(e);
    return VBNMF_OK;
}

}  // namespace

extern "C" {

int32_t vbnmf_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cnt;
}

void vbnmf_engine_destroy(vbnmf_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->poisoned) {
        // A wait timed out earlier.  If the streams have drained since, destroy as usual; if work is still queued (a
        // collective whose peer is gone), waiting would hang and freeing would pull memory from under queued kernels:
        // the handle is abandoned -- the process is expected to end with the error it was given.
        const bool busy = (e->stream && hipStreamQuery(e->stream) == hipErrorNotReady) ||
                          (e->cstream && hipStreamQuery(e->cstream) == hipErrorNotReady);
        (void)hipGetLastError();
        if (busy) {
            if (e->comm) for (vbnmf_engine *&q : e->comm->members) if (q == e) q = nullptr;
            return;
        }
    }
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->cstream) (void)hipStreamSynchronize(e->cstream);
    if (e->comm) {
        for (vbnmf_engine *&q : e->comm->members) if (q == e) q = nullptr;
        e->comm->tables_ready = false;
    }
    free_side(e->A); free_side(e->B);
    dev_free(e->lw); dev_free(e->llw); dev_free(e->ew); dev_free(e->dw);
    dev_free(e->lh); dev_free(e->llh); dev_free(e->eh); dev_free(e->dh);
    if (!e->epart_in_red) dev_free(e->epart);
    dev_free(e->bpW); dev_free(e->bpH); dev_free(e->csl); dev_free(e->csum); dev_free(e->upd_tab);
    dev_free(e->d_perm); dev_free(e->d_ids[0]); dev_free(e->d_ids[1]); dev_free(e->d_table); dev_free(e->svd_ws); dev_free(e->svd_status);
    dev_free(e->red); dev_free(e->red_g); dev_free(e->d_out);
    for (hipEvent_t ev : e->ev_ring) (void)hipEventDestroy(ev);
    if (e->cstream) (void)hipStreamDestroy(e->cstream); dev_free(e->dbg); dev_free(e->logtab); dev_free(e->ctl);
    dev_free(e->ctl2); dev_free(e->bpW_alt); dev_free(e->bpH_alt);
    release_stage(e);
    if (e->h_out) (void)hipHostFree(e->h_out);
    if (e->h_hist) (void)hipHostFree(e->h_hist);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev2) (void)hipEventDestroy(e->ev2);
    if (e->ev3) (void)hipEventDestroy(e->ev3);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// geometry_rank: the rank whose LDS row size the tiled layouts are cut for (>= r: several ranks of a sweep share one pair
// of layouts); 0: this rank's class in the matrix's plan (vbnmf_matrix_plan_ranks), its own geometry without one.
int vbnmf_device_warmup(int32_t device);

int vbnmf_engine_create_geom(const vbnmf_matrix *X, int64_t cb, int64_t ce, int64_t m_global, int32_t r, int32_t geometry_rank,
                             int32_t device, vbnmf_engine **out)
{
    if (!out) return fail(VBNMF_ERR_BAD_ARG, "out pointer is NULL");
    *out = nullptr;
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (r < 1 || r > VBNMF_MAX_RANK) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [1, %d]", r, VBNMF_MAX_RANK);
    if (geometry_rank != 0 && (geometry_rank < r || geometry_rank > VBNMF_MAX_RANK))
        return fail(VBNMF_ERR_BAD_ARG, "geometry rank %d cannot serve rank %d (it must lie in [rank, %d])", geometry_rank, r, VBNMF_MAX_RANK);
    if (cb < 0 || ce > X->M.m || cb >= ce) return fail(VBNMF_ERR_BAD_ARG, "column range [%lld, %lld) is outside the matrix", (long long)cb, (long long)ce);
    if (m_global < ce - cb) return fail(VBNMF_ERR_BAD_ARG, "m_global is smaller than the partition");
    if (X->M.shell && !(cb == 0 && ce == X->M.m))
        return fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries to cut a partition's layout from");
    if (int rc = check_device(device)) return rc;
    HIPCHECK(hipSetDevice(device));

    vbnmf_engine *e = new (std::nothrow) vbnmf_engine();
    if (!e) return fail(VBNMF_ERR_OOM, "out of host memory");
    // VBNMF_BUILD_TIMES=1: where the seconds of this creation go (offsets from here; the layouts' own phases come from host.cpp)
    const bool tl_on = getenv("VBNMF_BUILD_TIMES") != nullptr;
    const auto tl_0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (tl_on) fprintf(stderr, "  engine create +%.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tl_0).count(), what);
    };
    e->device = device;
    e->n = X->M.n; e->m = ce - cb; e->m_global = m_global; e->col_begin = cb;
    e->r = r; e->R = padded_rank(r);
    if (tl_pad_R > e->R) e->R = tl_pad_R;                    // (engines of several ranks meant for ONE batch: vbnmf_set_engine_padding)
    e->NT = sweep_threads(e->R);
    {
        // Blocks of the update kernels.  Fewer than one per CU on small matrices (one pass of the longer factor per block)
        // was measured and is SLOWER: 200 x 500 at rank 3 36.1 against 30.9 us per step with 2 blocks, 2 000 x 10 000 at
        // rank 5 82.7 against 79.2 with 59 -- the gather of the task partials is address work (64 scattered 8-byte loads
        // per wave instruction) that 256 CUs' texture units share and 2 do not.  VBNMF_UPDATE_BLOCKS overrides (experiments).
        int ub = kUpdateBlocks;
        if (const char *sv = getenv("VBNMF_UPDATE_BLOCKS")) { int v = atoi(sv); if (v >= 1 && v <= kUpdateBlocks) ub = v; }
        if (tl_grid_ub > 0) ub = std::min(ub, tl_grid_ub);          // (engines meant for a batch: vbnmf_set_engine_grid)
        e->ub = ub;
    }
    e->wide = !X->M.counts_int;
    e->partitioned = (ce - cb) != m_global;
    if (int rc = sweep_workgroups(device, e->partitioned, e->n_wg)) { delete e; return rc; }
    if (!e->partitioned && tl_grid_nwg > 0) e->n_wg = std::min(e->n_wg, tl_grid_nwg);
    int rc = VBNMF_OK;
    // This process's FIRST use of the device (HIP context, first allocation, code object: 0.1-0.15 s, once) runs on a helper
    // thread beside the host's cut of the layouts and is waited for where the first upload needs it (a failure there
    // resurfaces at that upload).  Later engines find the device warm and start no thread.
    static std::atomic<bool> warm_started[16];
    std::thread warm;
    std::mutex warm_mu;
    if (device < 16 && !warm_started[device].exchange(true)) warm = std::thread([device] { (void)vbnmf_device_warmup(device); });
    auto join_warm = [&] { std::lock_guard<std::mutex> g(warm_mu); if (warm.joinable()) warm.join(); };
    auto bail = [&](int code) { join_warm(); vbnmf_engine_destroy(e); return code; };

    try {
        std::vector<int32_t> part_order;                         // a partition orders its own cells (both sides alike)
        if (!(cb == 0 && ce == X->M.m)) part_order = compute_cell_order(X->M, cb, ce);
        std::shared_ptr<const Layout> shared2[2];
        Layout own2[2];
        const Layout *Ls[2] = {nullptr, nullptr};
        // The two sides are cut (or taken from the matrix's cache) and uploaded SIDE BY SIDE: the cell side on a second host
        // thread, the gene side here.  Each cut is memory-bound well before it uses all host threads, and a side's upload
        // (200 MB of pageable memory at the headline size) runs beside the other side's cut.  Engine creation at the headline
        // size: 0.41 -> 0.3 s on the GPU box (profiles/r05_setup_times.txt).  VBNMF_SERIAL_SIDES=1: one after the other.
        mark("start");
        if (cb == 0 && ce == X->M.m && !X->M.shell) (void)X->M.cell_order();          // (both sides start from it: formed once, here)
        mark("cell order");
        auto do_side = [&](int side) -> int {
            int src = VBNMF_OK;
            const int64_t nmaj = side == 0 ? e->n : e->m, nmin = side == 0 ? e->m : e->n;
            // The geometry is that of the matrix's rank CLASS (vbnmf_matrix_plan_ranks; without a plan the class is this
            // rank's own): the ranks of a sweep share one pair of layouts, cut for the widest rows among them.
            const int Rc = std::max(e->R, geometry_rank ? padded_rank(geometry_rank) : plan_class(X, e->R));
            const int64_t nnz_part = (cb == 0 && ce == X->M.m) ? X->M.nnz : X->M.colptr[ce] - X->M.colptr[cb];
            LayoutParams lp = default_layout_params(nmaj, nmin, Rc, e->n_wg, nnz_part);
            std::shared_ptr<const Layout> &shared = shared2[side];
            Layout &own = own2[side];
            const Layout *L = &own;
            if (cb == 0 && ce == X->M.m) {                   // whole matrix: the layout may already exist (another rank, a restart)
                shared = shared_layout(X, side, lp, src);
                L = shared.get();
            } else {
                src = build_layout(X->M, cb, ce, side, lp, &part_order, own);
            }
            mark(side == 0 ? "gene side cut" : "cell side cut");
            join_warm();
            if (!src) src = upload_side(*L, e->R, device, shared ? X : nullptr, side == 0 ? e->A : e->B);
            mark(side == 0 ? "gene side uploaded" : "cell side uploaded");
            Ls[side] = L;
            return src;
        };
        static const bool serial_sides = [] { const char *v = getenv("VBNMF_SERIAL_SIDES"); return v && v[0] == '1'; }();
        int rc1 = VBNMF_OK;
        std::string msg1;
        bool oom1 = false;
        if (serial_sides) {
            rc = do_side(0);
            if (!rc) rc = do_side(1);
        } else {
            static const int side_share = [] { const char *v = getenv("VBNMF_SIDE_THREAD_SHARE"); return v ? std::max(1, atoi(v)) : 1; }();
            std::thread cell_side([&] {
                try {
                    if (hipSetDevice(device) != hipSuccess) { rc1 = VBNMF_ERR_HIP; msg1 = "hipSetDevice failed on the layout thread"; return; }
                    set_thread_share(side_share);
                    rc1 = do_side(1);
                    if (rc1) msg1 = last_error_cstr();               // (error messages are per host thread)
                } catch (const std::bad_alloc &) { oom1 = true; }
            });
            set_thread_share(side_share);
            try { rc = do_side(0); } catch (...) { set_thread_share(1); cell_side.join(); throw; }
            set_thread_share(1);
            cell_side.join();
            if (oom1) throw std::bad_alloc();
            if (!rc && rc1) rc = fail(rc1, "%s", msg1.c_str());
        }
        if (!rc) { e->nnz = Ls[0]->nnz; e->cell_perm = Ls[0]->cell_perm; }
        if (!rc && Ls[1]->cell_perm != e->cell_perm) rc = fail(VBNMF_ERR_STATE, "the two sides' layouts disagree on the order of the cells");
        if (!rc) {
            // One launch for both posterior updates (k_update2): unpartitioned engines whose blocks' work fits the kernel's
            // table; VBNMF_NO_UPDATE_PAIR=1 keeps the two launches (A/B switch, and the form the pair is held to in
            // tests/test_gpu_update_pair.py).
            // Where it pays (round 5, same-box A/Bs in profiles/r05_pair_ab*.txt, r05_small_pair_ab.txt): the launch it saves is
            // worth ~3 us of a step whatever the size, the column sums it needs cost the gene side of the sweep ~60 + 4 R
            // instructions per SLICE.  5 000 x 20 000 at rank 8: 64.0 -> 60.1 us per step (+6 %); C2 (2 000 x 10 000 dense,
            // rank 5) +1.9 %; C1 and the PBMC-sized sample +-0; the headline (4.9e7 entries, rank 10) -0.3 %, rank 20 -1.2 %.
            // So: on up to 2.5e8 entries x padded rank, off beyond.  VBNMF_UPDATE_PAIR=1 / VBNMF_NO_UPDATE_PAIR=1 force it.
            const char *np = getenv("VBNMF_NO_UPDATE_PAIR"), *fp = getenv("VBNMF_UPDATE_PAIR");
            const bool forced = fp && fp[0] == '1';
            const bool pays = (double)Ls[0]->nnz * (double)e->R <= 2.5e8;
            e->pair = !e->partitioned && !(np && np[0] == '1') && (forced || pays) && Ls[0]->n_wg == Ls[1]->n_wg;
            if (e->pair) {
                std::vector<uint32_t> tab;
                e->pair = build_update_table(*Ls[0], *Ls[1], e->ub, e->R, tab, e->upd_stride4, e->upd_V, e->upd_ids_off);
                if (e->pair) {
                    uint32_t *d = nullptr;
                    rc = dev_upload(&d, tab);
                    e->upd_tab = reinterpret_cast<uint4 *>(d);
                }
            }
        }
    } catch (const std::bad_alloc &) {
        rc = fail(VBNMF_ERR_OOM, "out of host memory building the tiled layout");
    }
    join_warm();
    if (rc) return bail(rc);
    mark("both sides, update table");
    {
        // The engine's stream, before the first fill: everything that initialises the engine's own buffers below is queued ON it
        // (hipMemsetAsync), so it is ordered with the engine's kernels.  A plain hipMemset goes to the device's null stream,
        // which this non-blocking stream does not wait for, and may return before the fill has run: with several host threads
        // creating engines side by side (vb_factorize(concurrent=K)) a fill queued behind the other threads' null-stream work
        // could land AFTER the engine's first kernels had written the buffer -- zeroed block partials, a different trajectory
        // (seen once in round 5, tests/test_gpu_end_to_end.py::test_concurrent_units_give_the_same_result).
        // (Created HERE and not at the top: this process's first use of the device -- 0.1 s of HIP start-up -- then happens on the
        // cell side's thread, at its upload, beside the gene side's cut, instead of in front of both.)
        hipError_t hs;
        if ((hs = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess ||
            (hs = hipEventCreate(&e->ev0)) != hipSuccess || (hs = hipEventCreate(&e->ev1)) != hipSuccess ||
            (hs = hipEventCreate(&e->ev2)) != hipSuccess || (hs = hipEventCreate(&e->ev3)) != hipSuccess)
            return bail(fail(VBNMF_ERR_HIP, "engine setup failed: %s", hipGetErrorString(hs)));
        e->own_stream = true;
    }

    if (!e->cell_perm.empty() && (rc = dev_upload(&e->d_perm, e->cell_perm))) return bail(rc);
    e->lgx = (cb == 0 && ce == X->M.m) ? X->lgx : sum_lgamma_x1(X->M, cb, ce);
    if (cb == 0 && ce == X->M.m) {                    // whole matrix: formed once per matrix, not per engine (a pass over X)
        std::call_once(X->xlx_once, [&] { X->xlx = sum_xlogx(X->M, 0, X->M.m); });      // (a shell carries the value already)
        e->xlx = X->xlx;
    } else {
        e->xlx = sum_xlogx(X->M, cb, ce);
    }
    {
        // Cache policy of the entry stream (kernels.h: ld_stream).  What a step moves between two uses of a line: both sides' entry
        // streams and the per-task partial rows, written and read back.  Beyond the Infinity Cache (256 MB: the headline moves
        // 410 + 2 x 75 MB) the stream is read non-temporally so that the partial rows and the state stay on the die; inside it
        // (C2: 130 MB; 5 000 x 20 000: 61 MB) the stream itself stays resident from step to step and the default policy is the
        // faster one (profiles/r05_nt_ab.txt, r05_small_nt_ab.txt).  VBNMF_STREAM_NT=0 / 1 forces it.
        const double entry_b = e->wide ? 12.0 : 4.0;
        const double moved = entry_b * ((double)e->A.n_slots + (double)e->B.n_slots) +
                             2.0 * 8.0 * e->R * 64.0 * ((double)e->A.n_slices + (double)e->B.n_slices);
        e->stream_nt = moved > 200e6;
        if (const char *v = getenv("VBNMF_STREAM_NT")) e->stream_nt = v[0] == '1';
    }
    mark("sum x log x");
    static_assert(kLdsRowBase == kLdsReserveBytes, "host and device disagree on the sweep's LDS reserve");
    e->lds_bytes = kLdsRowBase + std::max((size_t)e->A.block_width * e->A.row_slots, (size_t)e->B.block_width * e->B.row_slots) * 16;
    if (e->lds_bytes > 160 * 1024) return bail(fail(VBNMF_ERR_BAD_ARG, "the layout's blocks need %zu bytes of LDS", e->lds_bytes));
    {
        std::vector<LogTabEntry> tab(kLogTabSize);
        fill_log_table(tab.data());
        if ((rc = dev_upload(&e->logtab, tab))) return bail(rc);
        if ((rc = dev_alloc(&e->ctl, 1))) return bail(rc);
        const char *nf = getenv("VBNMF_NO_CONTROL_FOLD");
        // (partitioned engines: the evidence partials of the two sweeps must fit the fixed slots of the second all-reduce)
        e->fold = !(nf && nf[0] == '1') && (!e->partitioned || 2 * (int64_t)e->n_wg <= kEvSlots);
        if (e->fold && (rc = dev_alloc(&e->ctl2, 2))) return bail(rc);
        if (e->fold || e->pair) {
            if ((rc = dev_alloc(&e->bpW_alt, (size_t)kUpdateBlocks * (e->R + 2))) ||
                (rc = dev_alloc(&e->bpH_alt, (size_t)kUpdateBlocks * (e->R + 2)))) return bail(rc);
            if (hipMemsetAsync(e->bpW_alt, 0, (size_t)kUpdateBlocks * (e->R + 2) * sizeof(double), e->stream) != hipSuccess ||
                hipMemsetAsync(e->bpH_alt, 0, (size_t)kUpdateBlocks * (e->R + 2) * sizeof(double), e->stream) != hipSuccess) return bail(fail(VBNMF_ERR_HIP, "hipMemsetAsync failed"));
        }
        if (e->pair) {
            if ((rc = dev_alloc(&e->csl, (size_t)std::max<int64_t>(e->A.n_slices, 1) * e->R)) || (rc = dev_alloc(&e->csum, (size_t)e->n_wg * e->R))) return bail(rc);
            if (hipMemsetAsync(e->csum, 0, (size_t)e->n_wg * e->R * sizeof(double), e->stream) != hipSuccess) return bail(fail(VBNMF_ERR_HIP, "hipMemsetAsync failed"));
        }
    }

    const size_t nR = (size_t)e->n * e->R, mR = (size_t)e->m * e->R;
    const size_t bpn = (size_t)kUpdateBlocks * (e->R + 2);
    e->red_count = (int64_t)nR + e->R + 4;
    if ((rc = dev_alloc(&e->lw, nR)) || (rc = dev_alloc(&e->llw, nR)) || (rc = dev_alloc(&e->ew, nR)) || (rc = dev_alloc(&e->dw, nR)) ||
        (rc = dev_alloc(&e->lh, mR)) || (rc = dev_alloc(&e->llh, mR)) || (rc = dev_alloc(&e->eh, mR)) || (rc = dev_alloc(&e->dh, mR)) ||
        (rc = dev_alloc(&e->bpW, bpn)) || (rc = dev_alloc(&e->bpH, bpn)) ||
        (rc = dev_alloc(&e->red, (size_t)red_alloc_count(e))) || (rc = dev_alloc(&e->d_out, 8)))
        return bail(rc);
    if (e->partitioned && 2 * (int64_t)e->n_wg <= kEvSlots) { e->epart = e->red + e->red_count; e->epart_in_red = true; }
    else if ((rc = dev_alloc(&e->epart, 2 * (size_t)e->n_wg))) return bail(rc);
    hipError_t he;
    if ((he = hipHostMalloc((void **)&e->h_out, kHostOut * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (he = hipHostGetDevicePointer((void **)&e->h_out_dev, e->h_out, 0)) != hipSuccess)
        return bail(fail(VBNMF_ERR_HIP, "engine setup failed: %s", hipGetErrorString(he)));
    std::memset(e->h_out, 0, kHostOut * sizeof(double));
    if (getenv("VBNMF_DEBUG_TIMES")) {
        e->dbg_count = 2 * (size_t)e->n_wg * (2 + 2 * (e->NT / 64));
        if ((rc = dev_alloc(&e->dbg, e->dbg_count))) return bail(rc);
    }
    if ((he = hipMemsetAsync(e->ew, 0, nR * sizeof(double), e->stream)) != hipSuccess || (he = hipMemsetAsync(e->dw, 0, nR * sizeof(double), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->dh, 0, mR * sizeof(double), e->stream)) != hipSuccess || (he = hipMemsetAsync(e->bpW, 0, bpn * sizeof(double), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->bpH, 0, bpn * sizeof(double), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->red, 0, (size_t)red_alloc_count(e) * sizeof(double), e->stream)) != hipSuccess)
        return bail(fail(VBNMF_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(he)));
    // (sum lgamma(x + 1) into its slot of the reduce buffer: BEHIND the fill above, on the same stream; e->lgx lives as long as the engine)
    if (e->epart_in_red && (he = hipMemcpyAsync(e->red + e->red_count + kEvSlots, &e->lgx, sizeof(double), hipMemcpyHostToDevice, e->stream)) != hipSuccess)
        return bail(fail(VBNMF_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)));
    // creation ends with the engine's buffers in their initial state whatever else the device is doing
    if ((he = hipStreamSynchronize(e->stream)) != hipSuccess) return bail(fail(VBNMF_ERR_HIP, "engine setup failed: %s", hipGetErrorString(he)));
    mark("state arrays, initial fills, done");
    *out = e;
    return VBNMF_OK;
}

int vbnmf_engine_create_part(const vbnmf_matrix *X, int64_t cb, int64_t ce, int64_t m_global, int32_t r,
                             int32_t device, vbnmf_engine **out)
{
    return vbnmf_engine_create_geom(X, cb, ce, m_global, r, 0, device, out);
}

int vbnmf_matrix_preload_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg, int32_t device)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (side != 0 && side != 1) return fail(VBNMF_ERR_BAD_ARG, "side must be 0 or 1");
    if (geometry_rank < 1 || geometry_rank > VBNMF_MAX_RANK || n_wg < 1) return fail(VBNMF_ERR_BAD_ARG, "bad geometry");
    if (int rc = check_device(device)) return rc;
    HIPCHECK(hipSetDevice(device));
    const int R = padded_rank(geometry_rank);
    const int64_t nmaj = side == 0 ? X->M.n : X->M.m, nmin = side == 0 ? X->M.m : X->M.n;
    int rc = VBNMF_OK;
    try {
        const LayoutParams lp = default_layout_params(nmaj, nmin, R, n_wg, X->M.nnz);
        std::shared_ptr<const Layout> L = shared_layout(X, side, lp, rc);
        if (rc) return rc;
        DeviceSide S;                                // the shared arrays stay in the matrix's cache; the per-engine part goes
        rc = upload_side(*L, R, device, X, S, false);
        free_side(S);
    } catch (const std::bad_alloc &) {
        rc = fail(VBNMF_ERR_OOM, "out of host memory building the tiled layout");
    }
    return rc;
}

// First use of a device by this process, ahead of need: the HIP context, the first allocation and the load of the
// library's code object cost ~0.15 s the first time and nothing afterwards.  A process that has to wait for something else
// first (a sharded sweep's peers waiting for the layouts) spends that wait here.
int vbnmf_device_warmup(int32_t device)
{
    if (int rc = check_device(device)) return rc;
    HIPCHECK(hipSetDevice(device));
    double *p = nullptr;
    if (int rc = dev_alloc(&p, 1024)) return rc;
    double host[8] = {0};
    hipError_t he = hipMemcpy(p, host, sizeof host, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(k_test_special, dim3(1), dim3(64), 0, nullptr, 3, (int64_t)0, (const double *)p, p, (const LogTabEntry *)nullptr);   // n = 0: loads the module, touches nothing
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipDeviceSynchronize();
    dev_free(p);
    if (he != hipSuccess) return fail(VBNMF_ERR_HIP, "device warm-up failed: %s", hipGetErrorString(he));
    return VBNMF_OK;
}

int vbnmf_device_sweep_workgroups(int32_t device, int32_t *n_wg)
{
    if (!n_wg) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (int rc = check_device(device)) return rc;
    int v = 0;
    if (int rc = sweep_workgroups(device, false, v)) return rc;
    *n_wg = v;
    return VBNMF_OK;
}

int vbnmf_engine_create(const vbnmf_matrix *X, int32_t r, int32_t device, vbnmf_engine **out)
{
    if (!X) { if (out) *out = nullptr; return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL"); }
    return vbnmf_engine_create_part(X, 0, X->M.m, X->M.m, r, device, out);
}

int vbnmf_engine_dims(const vbnmf_engine *e, int64_t *n, int64_t *m_local, int32_t *r)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (n) *n = e->n;
    if (m_local) *m_local = e->m;
    if (r) *r = e->r;
    return VBNMF_OK;
}

int vbnmf_engine_get_stream(vbnmf_engine *e, void **stream)
{
    if (!e || !stream) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    *stream = (void *)e->stream;
    return VBNMF_OK;
}

int vbnmf_engine_set_stream(vbnmf_engine *e, void *stream)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (int rc = use_device(e)) return rc;
    HIPCHECK(hipStreamSynchronize(e->stream));
    if (e->own_stream) { (void)hipStreamDestroy(e->stream); e->own_stream = false; }
    e->stream = (hipStream_t)stream;
    return VBNMF_OK;
}

// column-major n x r (R matrix) -> device [n][R]; or r x m column-major (already index-major) -> [m][R].
// perm (cell-indexed arrays only): device row p holds the caller's major perm[p] (the layout's order of the cells).
// host threads for a conversion of a factor's arrays: one per 32 K elements (a 200 x 3 factor is not worth waking anyone for)
static int state_threads(int64_t nmaj, int R)
{
    return (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (nmaj * (int64_t)R) >> 15));
}

static void to_index_major(const double *src, int64_t nmaj, int r, int R, bool src_is_major_contiguous, double *dst,
                           const std::vector<int32_t> *perm = nullptr)
{
    const int32_t *pm = (perm && !perm->empty()) ? perm->data() : nullptr;
    parallel_for(nmaj, [&](int64_t b, int64_t e, int) {
        for (int64_t M = b; M < e; M++) {
            const int64_t S = pm ? pm[M] : M;
            double *d = dst + (size_t)M * R;
            for (int k = 0; k < r; k++) d[k] = src_is_major_contiguous ? src[(size_t)S * r + k] : src[S + (size_t)k * nmaj];
            for (int k = r; k < R; k++) d[k] = 0.0;
        }
    }, state_threads(nmaj, R));
}

static void to_index_major(const double *src, int64_t nmaj, int r, int R, bool src_is_major_contiguous, std::vector<double> &dst,
                           const std::vector<int32_t> *perm = nullptr)
{
    dst.resize((size_t)nmaj * R);
    to_index_major(src, nmaj, r, R, src_is_major_contiguous, dst.data(), perm);
}

static void from_index_major(const double *src, int64_t nmaj, int r, int R, bool dst_is_major_contiguous, double *dst,
                             const std::vector<int32_t> *perm = nullptr)
{
    const int32_t *pm = (perm && !perm->empty()) ? perm->data() : nullptr;
    parallel_for(nmaj, [&](int64_t b, int64_t e, int) {
        for (int64_t M = b; M < e; M++) {
            const int64_t D = pm ? pm[M] : M;
            for (int k = 0; k < r; k++) {
                double v = src[(size_t)M * R + k];
                if (dst_is_major_contiguous) dst[(size_t)D * r + k] = v; else dst[D + (size_t)k * nmaj] = v;
            }
        }
    }, state_threads(nmaj, R));
}

static void from_index_major(const std::vector<double> &src, int64_t nmaj, int r, int R, bool dst_is_major_contiguous, double *dst,
                             const std::vector<int32_t> *perm = nullptr)
{
    from_index_major(src.data(), nmaj, r, R, dst_is_major_contiguous, dst, perm);
}

int vbnmf_engine_set_state(vbnmf_engine *e, const double *lw, const double *lh, const double *eh)
{
    if (!e || !lw || !lh || !eh) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (int rc = use_device(e)) return rc;
    if (e->prime_pending) {
        // A partitioned engine's previous set_state returned with its copies out of the pinned staging buffer still queued
        // (state_finish waits for them): rewriting that buffer -- or handing it to the pool when it grows -- under in-flight
        // DMA would load a corrupted state without any error.  Wait for the stream first (bounded: a state exchange whose peer
        // never joined sits on it).
        if (int rc = bounded_stream_sync(e, e->stream, "the previous set_state of this engine")) return rc;
    }
    e->has_state = false; e->stats_ready = false; e->step_pending = false; e->prime_pending = false;
    e->ml_ready = false;
    {
        // the three arrays in index-major form in ONE pinned staging buffer, three asynchronous copies, one wait
        const size_t nR = (size_t)e->n * e->R, mR = (size_t)e->m * e->R;
        if (int rc = ensure_stage(e, nR + 2 * mR)) return rc;
        double *s_lw = e->h_stage, *s_lh = s_lw + nR, *s_eh = s_lh + mR;
        to_index_major(lw, e->n, e->r, e->R, false, s_lw);
        HIPCHECK(hipMemcpyAsync(e->lw, s_lw, nR * sizeof(double), hipMemcpyHostToDevice, e->stream));
        to_index_major(lh, e->m, e->r, e->R, true, s_lh, &e->cell_perm);
        HIPCHECK(hipMemcpyAsync(e->lh, s_lh, mR * sizeof(double), hipMemcpyHostToDevice, e->stream));
        to_index_major(eh, e->m, e->r, e->R, true, s_eh, &e->cell_perm);
        HIPCHECK(hipMemcpyAsync(e->eh, s_eh, mR * sizeof(double), hipMemcpyHostToDevice, e->stream));
        // (prime_state's kernels follow on the same stream; its closing synchronise also covers the staging buffer)
    }
    return prime_state(e);
}

// The 'random' initialiser of vb_init (reference R/bayesian.R:111-115, 162-170) drawn on the device (init.h):
// lw = ew ~ Gamma(shape aw, scale bw / aw), lh = eh ~ Gamma(ah, bh / ah), dw = dh = 0, then the statistics of that state.
int vbnmf_engine_random_state(vbnmf_engine *e, double aw, double bw, double ah, double bh, uint64_t seed)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!(aw > 0.0) || !(bw > 0.0) || !(ah > 0.0) || !(bh > 0.0)) return fail(VBNMF_ERR_BAD_ARG, "Gamma shapes and means must be positive");
    if (int rc = use_device(e)) return rc;
    e->has_state = false; e->stats_ready = false; e->step_pending = false; e->prime_pending = false;
    e->ml_ready = false; e->ids_valid = false;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t nw = e->n * e->R, nh = e->m * e->R;
    hipLaunchKernelGGL(k_gamma_init, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, e->stream, e->lw, e->ew, e->dw, e->n, (int64_t)0, e->r, e->R, aw, bw, 0u, k0, k1, (const int32_t *)nullptr);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_gamma_init, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, e->stream, e->lh, e->eh, e->dh, e->m, e->col_begin, e->r, e->R, ah, bh, 1u, k0, k1, (const int32_t *)e->d_perm);
    HIPCHECK(hipGetLastError());
    return prime_state(e);
}

int vbnmf_engine_state_finish(vbnmf_engine *e)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->prime_pending) return fail(VBNMF_ERR_STATE, "state_finish without a pending set_state");
    if (int rc = use_device(e)) return rc;
    if (e->partitioned) {                            // the state exchange sits on this stream: a peer that never joins it must not hang us
        if (int rc = bounded_stream_sync(e, e->stream, "the state exchange of a partitioned engine")) return rc;
    } else {
        HIPCHECK(hipStreamSynchronize(e->stream));
    }
    if (int rc = harvest_timing(e)) return rc;
    e->sweep_ms = 0.0; e->sweep_launches = 0;        // the priming sweep is not a step
    e->prime_pending = false;
    e->stats_ready = true;
    return VBNMF_OK;
}

int vbnmf_engine_step_local(vbnmf_engine *e, double aw, double bw, double ah, double bh, double fudge)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (int rc = use_device(e)) return rc;                                   // (first: an engine that timed out says so, whatever its state)
    if (!e->has_state || !e->stats_ready) return fail(VBNMF_ERR_STATE, "step before set_state (or before state_finish on a partitioned engine)");
    if (e->step_pending) return fail(VBNMF_ERR_STATE, "step_local called twice without step_finish");
    if (e->pair) {
        if (int rc = launch_update2(e, aw, bw, ah, bh, fudge)) return rc;
    } else {
        if (int rc = launch_update(e, true, aw, bw, fudge)) return rc;
        if (int rc = launch_update(e, false, ah, bh, fudge)) return rc;
    }
    if (int rc = launch_sweep(e)) return rc;
    if (e->partitioned) { if (int rc = launch_pack(e)) return rc; }
    e->step_pending = true;
    return VBNMF_OK;
}

int vbnmf_engine_reduce_buffer(vbnmf_engine *e, void **device_ptr, int64_t *count)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (device_ptr) *device_ptr = e->red;
    if (count) *count = e->red_count;
    return VBNMF_OK;
}

int vbnmf_engine_step_finish(vbnmf_engine *e, double *lkh, double *stats)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (int rc = use_device(e)) return rc;
    if (!e->step_pending) return fail(VBNMF_ERR_STATE, "step_finish without step_local");
    if (int rc = launch_final(e)) return rc;
    if (int rc = wait_result(e)) return rc;
    e->step_pending = false;
    if (int rc = harvest_timing(e)) return rc;
    if (lkh) *lkh = e->h_out[0];
    if (stats) for (int q = 0; q < 4; q++) stats[q] = e->h_out[1 + q];
    return VBNMF_OK;
}

int vbnmf_engine_step(vbnmf_engine *e, double aw, double bw, double ah, double bh, double fudge, double *lkh, double *stats)
{
    if (e && e->partitioned) return fail(VBNMF_ERR_STATE, "a partitioned engine needs step_local / all-reduce / step_finish");
    if (int rc = vbnmf_engine_step_local(e, aw, bw, ah, bh, fudge)) return rc;
    return vbnmf_engine_step_finish(e, lkh, stats);
}

// ---------------------------------------------------------------- device-driven loops
}  // extern "C"

namespace {

// Pinned, device-visible history of a device-driven loop (k_control / k_ml_control write one row per step straight
// into host memory); kept by the engine and grown on demand, so a run allocates nothing.
int ensure_history(vbnmf_engine *e, size_t doubles)
{
    if (doubles <= e->h_hist_count) return VBNMF_OK;
    if (e->h_hist) {
        HIPCHECK(hipStreamSynchronize(e->stream));
        (void)hipHostFree(e->h_hist);
        e->h_hist = nullptr; e->h_hist_dev = nullptr; e->h_hist_count = 0;
    }
    const size_t want = std::max<size_t>(doubles, 9 * 1024);
    hipError_t he = hipHostMalloc((void **)&e->h_hist, want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
    if (he == hipSuccess) he = hipHostGetDevicePointer((void **)&e->h_hist_dev, e->h_hist, 0);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        if (e->h_hist) { (void)hipHostFree(e->h_hist); e->h_hist = nullptr; }
        return fail(VBNMF_ERR_OOM, "pinned history buffer (%zu bytes): %s", want * sizeof(double), hipGetErrorString(he));
    }
    e->h_hist_count = want;
    return VBNMF_OK;
}

// What a partitioned engine needs beside its own stream: the stream the all-reduces go on, the receive buffer, and
// a ring of events (five per step, steps queued at most three batches of eight ahead: 160 never wraps onto a pending one).
int ensure_comm_resources(vbnmf_engine *e)
{
    if (!e->cstream) HIPCHECK(hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking));
    if (!e->red_g) {
        if (int rc = dev_alloc(&e->red_g, (size_t)red_alloc_count(e))) return rc;
        HIPCHECK(hipMemsetAsync(e->red_g, 0, (size_t)red_alloc_count(e) * sizeof(double), e->stream));      // (ordered with the engine's kernels; see engine creation)
        HIPCHECK(hipStreamSynchronize(e->stream));
    }
    while (e->ev_ring.size() < 160) {
        hipEvent_t ev;
        HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->ev_ring.push_back(ev);
    }
    return VBNMF_OK;
}

hipEvent_t next_event(vbnmf_engine *e)
{
    hipEvent_t ev = e->ev_ring[e->ev_next];
    e->ev_next = (e->ev_next + 1) % e->ev_ring.size();
    return ev;
}

int rccl_check(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return VBNMF_OK;
    RcclApi &api = rccl_api();
    return fail(VBNMF_ERR_HIP, "%s failed: %s", what, api.GetErrorString ? api.GetErrorString(r) : "RCCL error");
}

int launch_control(vbnmf_engine *e, double *hist_dev, bool reduced)
{
    const int64_t nep = 2 * (int64_t)e->n_wg;
    const double *tail = reduced ? e->red_g + (size_t)e->n * e->R : nullptr;
    const double *small = reduced ? e->red_g + (size_t)e->n * e->R + e->R + 2 : nullptr;
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_control<RR>), dim3(1), dim3(1024), 0, e->stream, e->bpW, e->bpH, e->ub, e->epart, nep, e->lgx, e->r, (double)e->n, (double)e->m_global, e->ctl, hist_dev, e->h_out_dev, tail, small); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// The engines one device-driven loop advances together: a single engine, or the partition engines of a local group.
struct LoopGroup {
    vbnmf_engine **e;
    int count;
    vbnmf_comm *comm;      // null: one unpartitioned engine
};

// Device pointer tables of a local group's send / receive buffers (built once all members are attached).
int group_tables(vbnmf_comm *c)
{
    if (c->tables_ready) return VBNMF_OK;
    const int P = (int)c->members.size();
    std::vector<const double *> sb(P), ss(P);
    std::vector<double *> rb(P), rs(P);
    for (int p = 0; p < P; p++) {
        vbnmf_engine *e = c->members[p];
        if (int rc = ensure_comm_resources(e)) return rc;
        // second exchange of a device-driven step: the evidence slots behind the reduce buffer (control step folded into
        // the next update), or the two doubles k_tail_data forms (VBNMF_NO_CONTROL_FOLD=1)
        const size_t off = e->fold ? (size_t)e->red_count : (size_t)e->n * e->R + e->R + 2;
        sb[p] = e->red; rb[p] = e->red_g; ss[p] = e->red + off; rs[p] = e->red_g + off;
    }
    dev_free(c->d_send_big); dev_free(c->d_send_small); dev_free(c->d_recv_big); dev_free(c->d_recv_small);
    c->d_send_big = c->d_send_small = nullptr; c->d_recv_big = c->d_recv_small = nullptr;      // (tables of an earlier membership)
    if (int rc = dev_alloc(&c->d_send_big, (size_t)P)) return rc;
    if (int rc = dev_alloc(&c->d_send_small, (size_t)P)) return rc;
    if (int rc = dev_alloc(&c->d_recv_big, (size_t)P)) return rc;
    if (int rc = dev_alloc(&c->d_recv_small, (size_t)P)) return rc;
    HIPCHECK(hipMemcpy(c->d_send_big, sb.data(), P * sizeof(void *), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->d_send_small, ss.data(), P * sizeof(void *), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->d_recv_big, rb.data(), P * sizeof(void *), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(c->d_recv_small, rs.data(), P * sizeof(void *), hipMemcpyHostToDevice));
    c->tables_ready = true;
    return VBNMF_OK;
}

// One step of the device-driven VB loop, queued on every engine of the group.
//   unpartitioned : k_update(W, + the control step of the previous sweep) k_update(H) k_sweep(both sides)
//   partitioned   : k_update(W <- reduced statistics, + the control step of the previous sweeps) k_update(H) | gene-side sweep
//                   | k_pack_tail (main stream for an RCCL rank, comm stream for a local group) | event
//                   -> comm stream: all-reduce [swsum | rowSum(eh) | 2 scalars]     (RCCL, or k_group_sum)
//                   || main stream: cell-side sweep
//                   -> main stream: all-reduce [evidence slots | sum lgamma(x+1)]   (behind an event after the first one)
// so the n*R-double exchange travels while the cell-side sweep runs and only the small one sits
// between the sweep and the next update.  (VBNMF_NO_CONTROL_FOLD=1: k_tail_data, a two-double exchange and k_control
// close the step instead.)  The all-reduces are out of place (red -> red_g): steps queued past the stop leave `red`
// untouched, so repeating them reproduces the same sums.
int queue_vb_step(const LoopGroup &G, double fudge, bool hist, int max_it)
{
    if (!G.comm && G.e[0]->fold) {
        // k_update(W, with the control step of the PREVIOUS sweep folded in)  k_update(H)  k_sweep ; behind the last step of
        // the run, the control step alone (kernels.h: ControlFold)
        vbnmf_engine *e = G.e[0];
        const int t = ++e->fold_step;                               // 1-based step of this run
        ControlFold f{};
        f.prev = e->ctl2 + ((t - 1) & 1); f.next = e->ctl2 + (t & 1);
        f.epart = e->epart; f.nepart = 2 * (int64_t)e->n_wg;
        f.lgx = e->lgx; f.n = (double)e->n; f.m_global = (double)e->m_global;
        f.history = hist ? e->h_hist_dev : nullptr; f.out_host = e->h_out_dev;
        f.do_control = t > 1 ? 1 : 0;
        int rc;
        if (e->pair) {
            // k_update2(W and H, with the control step of the previous sweep folded in)  k_sweep : two launches per step
            rc = launch_update2(e, 0, 0, 0, 0, fudge, nullptr, &f);
            e->stop_ptr = &f.next->stop;
        } else {
            f.bpW_prev = e->bpW;
            std::swap(e->bpW, e->bpW_alt);                          // this step's gene-side partials go to the other table
            rc = launch_update(e, true, 0, 0, fudge, nullptr, &f);
            e->stop_ptr = &f.next->stop;
            if (!rc) rc = launch_update(e, false, 0, 0, fudge, f.next);
        }
        if (!rc) rc = launch_sweep(e);
        if (!rc && t == max_it) {
            ControlFold g = f;
            g.prev = f.next; g.next = e->ctl2 + ((t + 1) & 1);
            g.bpW_prev = e->bpW;
            g.do_control = 1; g.control_only = 1;
            rc = launch_update(e, true, 0, 0, fudge, nullptr, &g);
        }
        return rc;
    }
    if (!G.comm) {
        vbnmf_engine *e = G.e[0];
        int rc;
        if (e->pair) rc = launch_update2(e, 0, 0, 0, 0, fudge, e->ctl);
        else {
            rc = launch_update(e, true, 0, 0, fudge, e->ctl);
            if (!rc) rc = launch_update(e, false, 0, 0, fudge, e->ctl);
        }
        if (!rc) rc = launch_sweep(e);
        if (!rc) rc = launch_control(e, hist ? e->h_hist_dev : nullptr, false);
        return rc;
    }
    vbnmf_comm *c = G.comm;
    const int P = G.count;
    vbnmf_engine *L = G.e[0];                                  // leader: owner of the comm stream used by a local group
    const int64_t nbig = L->n * L->R + L->R + 2;
    const bool fold = L->fold;
    static const bool small_on_comm = [] { const char *v = getenv("VBNMF_SMALL_ON_COMM"); return v && v[0] == '1'; }();   // (see the second exchange below)
    hipEvent_t evA[64], evB[64];
    ControlFold last{};                                        // (fold) the control-only launch behind step max_it
    for (int p = 0; p < P; p++) {
        vbnmf_engine *e = G.e[p];
        int rc;
        if (fold) {
            // k_update(W <- reduced statistics, with the control step of the PREVIOUS sweeps folded in: no k_control launch
            // and no k_tail_data between the second all-reduce and the next update -- that exchange carries the evidence
            // partials of the two sweeps as they are, element-wise, and the fold adds them up)
            const int t = ++e->fold_step;
            ControlFold f{};
            f.prev = e->ctl2 + ((t - 1) & 1); f.next = e->ctl2 + (t & 1);
            f.bpW_prev = e->bpW; f.nbW = e->ub;
            std::swap(e->bpW, e->bpW_alt);
            f.tail_in = e->red_g + (size_t)e->n * e->R;
            f.epart = e->red_g + e->red_count; f.nepart = kEvSlots;
            f.lgx_in = e->red_g + e->red_count + kEvSlots;
            f.lgx = 0.0; f.n = (double)e->n; f.m_global = (double)e->m_global;
            f.history = hist && p == 0 ? e->h_hist_dev : nullptr; f.out_host = e->h_out_dev;
            f.do_control = t > 1 ? 1 : 0;
            rc = launch_update(e, true, 0, 0, fudge, nullptr, &f);
            e->stop_ptr = &f.next->stop;
            if (!rc) rc = launch_update(e, false, 0, 0, fudge, f.next);
            if (t == max_it) {
                last = f;
                last.prev = f.next; last.next = e->ctl2 + ((t + 1) & 1);
                last.do_control = 1; last.control_only = 1;
            }
        } else {
            rc = launch_update(e, true, 0, 0, fudge, e->ctl);
            if (!rc) rc = launch_update(e, false, 0, 0, fudge, e->ctl);
        }
        if (!rc) rc = launch_vb_side(e, true);
        if (rc) return rc;
        const int32_t *stop = e->stop_ptr ? e->stop_ptr : &e->ctl->stop;
        // The gene-side partials are packed into the send buffer (k_pack_tail: a gather of 100 MB of task rows at C5, 26 us
        // with the chip to itself, + the cell side's column sums) on the MAIN stream, between the two sweeps: the all-reduce
        // of n * R doubles must travel WHILE the cell-side sweep runs, and a pack queued beside that sweep on the comm
        // stream is starved by its persistent workgroups (profiles/r04_c5_step_timeline.txt: 168 us, ending after the
        // sweep -- the collective would start when the sweep is over).  VBNMF_PACK_ON_MAIN=0 puts it on the comm stream
        // (A/B switch: round 4's first half ran it there; on the final build it is slower for a local group of eight as
        // well, 0.373-0.376 against 0.370-0.371 ms per partition step on the same box).
        static const bool pack_on_main = [] { const char *v = getenv("VBNMF_PACK_ON_MAIN"); return !(v && v[0] == '0'); }();
        hipStream_t ps = pack_on_main ? e->stream : e->cstream;
        if (!pack_on_main) {
            hipEvent_t evS = next_event(e);
            HIPCHECK(hipEventRecord(evS, e->stream));
            HIPCHECK(hipStreamWaitEvent(e->cstream, evS, 0));
        }
        const int64_t cnt = e->n * e->R;
        hipLaunchKernelGGL(k_pack_tail, dim3((unsigned)((cnt + 255) / 256) + 1), dim3(256), 0, ps, e->A.part, e->A.inv_ptr, e->A.inv_task, e->n, e->R, e->red,
                           e->bpH, e->ub, stop);
        HIPCHECK(hipGetLastError());
        evA[p] = next_event(e);
        HIPCHECK(hipEventRecord(evA[p], ps));
    }
    if (c->kind == 0) {
        HIPCHECK(hipStreamWaitEvent(L->cstream, evA[0], 0));   // (a no-op when k_pack / k_tail_h ran on the comm stream themselves)
        if (int rc = rccl_check(rccl_api().AllReduce(L->red, L->red_g, (size_t)nbig, ncclDouble, ncclSum, c->nc, L->cstream), "ncclAllReduce")) return rc;
    } else {
        for (int p = 0; p < P; p++) HIPCHECK(hipStreamWaitEvent(L->cstream, evA[p], 0));
        hipLaunchKernelGGL(k_group_sum, dim3((unsigned)((nbig + 255) / 256)), dim3(256), 0, L->cstream, c->d_send_big, c->d_recv_big, P, nbig);
        HIPCHECK(hipGetLastError());
    }
    for (int p = 0; p < P; p++) {
        vbnmf_engine *e = G.e[p];
        if (int rc = launch_vb_side(e, false)) return rc;
        if (!fold) {
            hipLaunchKernelGGL(k_tail_data, dim3(1), dim3(1024), 0, e->stream, e->epart, 2 * (int64_t)e->n_wg, e->lgx, e->red + nbig, &e->ctl->stop);
            HIPCHECK(hipGetLastError());
        }
        if (small_on_comm || p > 0) {                            // (an event costs the stream a barrier packet: only where another stream waits on it)
            evB[p] = next_event(e);
            HIPCHECK(hipEventRecord(evB[p], e->stream));
        }
    }
    // The second exchange sits between the cell-side sweep and the next update on every rank's critical path, so it is
    // issued on the leader's MAIN stream: no hand-over to the comm stream and back (two cross-queue waits, microseconds
    // each, for a collective of a few KB).  It follows the first exchange through an event recorded behind it on the comm
    // stream -- long satisfied by then -- so the two collectives of a step never overlap and every rank issues them in the
    // same order.  VBNMF_SMALL_ON_COMM=1 puts it back on the comm stream (A/B switch).
    hipStream_t ss = small_on_comm ? L->cstream : L->stream;
    if (!small_on_comm) {
        hipEvent_t evBig = next_event(L);
        HIPCHECK(hipEventRecord(evBig, L->cstream));
        HIPCHECK(hipStreamWaitEvent(L->stream, evBig, 0));
    }
    const bool need_evD = small_on_comm || P > 1;
    hipEvent_t evD = need_evD ? next_event(L) : nullptr;
    if (c->kind == 0) {
        if (small_on_comm) HIPCHECK(hipStreamWaitEvent(L->cstream, evB[0], 0));
        const int64_t off = fold ? L->red_count : nbig, cnt = fold ? kEvSlots + 1 : 2;
        if (int rc = rccl_check(rccl_api().AllReduce(L->red + off, L->red_g + off, (size_t)cnt, ncclDouble, ncclSum, c->nc, ss), "ncclAllReduce")) return rc;
    } else {
        for (int p = small_on_comm ? 0 : 1; p < P; p++) HIPCHECK(hipStreamWaitEvent(ss, evB[p], 0));
        const int64_t cnt = fold ? kEvSlots + 1 : 2;
        hipLaunchKernelGGL(k_group_sum, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ss, c->d_send_small, c->d_recv_small, P, cnt);
        HIPCHECK(hipGetLastError());
    }
    if (need_evD) HIPCHECK(hipEventRecord(evD, ss));
    for (int p = 0; p < P; p++) {
        vbnmf_engine *e = G.e[p];
        if (need_evD && e->stream != ss) HIPCHECK(hipStreamWaitEvent(e->stream, evD, 0));
        if (!fold) { if (int rc = launch_control(e, hist && p == 0 ? e->h_hist_dev : nullptr, true)) return rc; }
        else if (last.control_only) {                           // behind the last step of the run: the control step alone
            ControlFold g = last;
            g.prev = e->ctl2 + (e->fold_step & 1); g.next = e->ctl2 + ((e->fold_step + 1) & 1);
            g.bpW_prev = e->bpW; g.nbW = e->ub;
            g.tail_in = e->red_g + (size_t)e->n * e->R;
            g.epart = e->red_g + e->red_count; g.lgx_in = e->red_g + e->red_count + kEvSlots;
            g.history = hist && p == 0 ? e->h_hist_dev : nullptr; g.out_host = e->h_out_dev;
            if (int rc = launch_update(e, true, 0, 0, fudge, nullptr, &g)) return rc;
        }
    }
    return VBNMF_OK;
}

// Host side of a device-driven loop.  Steps are queued in batches of eight, two batches ahead of the device, and batch
// b + 2 is queued exactly when the loop has not stopped at or before the last step of batch b -- a function of the
// device's (replicated) decisions alone, never of when this host happened to look.  Every process of a partitioned run
// therefore queues the same number of steps, i.e. the same sequence of collectives.  The host reads the pinned result
// block of engine 0: [5] = steps done, [6] = reason (raised after [5]), [7] = steps done (raised last).
template <class QueueStep>
int drive_loop(vbnmf_engine *e0, int max_it, bool fold, QueueStep &&queue_step)
{
    volatile double *ho = e0->h_out;
    const int B = 8;
    int queued = 0;
    auto queue_batch = [&]() -> int {
        for (int q = 0; q < B && queued < max_it; q++, queued++)
            if (int rc = queue_step()) return rc;
        return VBNMF_OK;
    };
    // "The stream is idle, no stop was raised, and fewer steps are reported than a drained stream must show": something
    // was lost on the device.  With the control step folded into the NEXT step's update (ControlFold / MlFold) step t is
    // only reported by step t + 1's launch, so a fully drained, healthy stream shows queued - 1 until the closing
    // control-only launch behind step max_it has been queued.
    auto idle_check = [&]() -> int {
        hipError_t q = hipStreamQuery(e0->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(VBNMF_ERR_HIP, "the loop failed on the device: %s", hipGetErrorString(q));
        const int expect = queued - ((fold && queued < max_it) ? 1 : 0);
        if (q == hipSuccess && ho[6] == 0.0 && (int)ho[7] < expect) return fail(VBNMF_ERR_HIP, "the device went idle before the queued steps finished");
        return VBNMF_OK;
    };
    // test hook (tests/test_gpu_control_fold.py): drain the stream before every look, i.e. the host thread stalled between
    // queueing and polling -- the case the idle check must not mistake for a lost step
    const char *drain_env = getenv("VBNMF_TEST_DRAIN_BEFORE_POLL");
    const bool drain_first = drain_env && drain_env[0] == '1';
    if (int rc = queue_batch()) return rc;
    if (int rc = queue_batch()) return rc;
    const double limit = wait_timeout_s();
    for (int b = 0;; b++) {
        const int target = (int)std::min<int64_t>((int64_t)(b + 1) * B, max_it);
        const auto t0 = std::chrono::steady_clock::now();          // the bound is per batch of B steps
        if (drain_first) {
            (void)hipStreamSynchronize(e0->stream);
            if (int rc = idle_check()) return rc;
        }
        for (long spins = 1; ho[6] == 0.0 && (int)ho[7] < target; spins++) {
            if ((spins & 0xFFFF) == 0) {
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited > limit) {
                    e0->poisoned = true;
                    return fail(VBNMF_ERR_HIP, "timed out after %.1f s (VBNMF_WAIT_TIMEOUT_S) waiting for step %d of the device-driven loop; "
                                "the last completed step is %d (%d queued)", waited, target, (int)ho[7], queued);
                }
                if (int rc = idle_check()) return rc;
            }
        }
        if (ho[6] != 0.0 && (int)ho[5] <= target) break;        // stopped inside a batch that is complete
        if (target >= max_it) break;
        if (int rc = queue_batch()) return rc;
    }
    return VBNMF_OK;
}

int run_group(const LoopGroup &G, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0, int32_t dn,
              const int32_t *flags, int32_t *it_out, double *lk0_out, double *lkh_out, int32_t *reason_out,
              double *history, int64_t history_rows)
{
    if (!hyper || !flags) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (max_it < 1 || dn < 1) return fail(VBNMF_ERR_BAD_ARG, "max_it and dn must be >= 1");
    if (history && history_rows < max_it) return fail(VBNMF_ERR_BAD_ARG, "history needs max_it rows of 9 doubles");
    for (int p = 0; p < G.count; p++) {
        vbnmf_engine *e = G.e[p];
        if (int rc = use_device(e)) return rc;
        if (!e->has_state || !e->stats_ready) return fail(VBNMF_ERR_STATE, "run before set_state (or before the state exchange of a partitioned engine)");
        if (e->step_pending) return fail(VBNMF_ERR_STATE, "run between step_local and step_finish");
    }
    vbnmf_engine *e0 = G.e[0];
    if (int rc = use_device(e0)) return rc;
    if (history) { if (int rc = ensure_history(e0, (size_t)max_it * 9)) return rc; }

    LoopCtl c{};
    for (int q = 0; q < 4; q++) { c.hyper[q] = hyper[q]; c.flags[q] = flags[q] ? 1 : 0; }
    c.lk0 = 0.0;                                                   // :336
    c.tol = tol; c.max_it = max_it; c.n0 = n0; c.dn = dn;
    std::vector<bool> timing(G.count);
    for (int p = 0; p < G.count; p++) {
        vbnmf_engine *e = G.e[p];
        timing[p] = e->timing;
        e->timing = false;                                         // event pairs cannot follow launches queued ahead
        e->ev_recorded = false; e->ev2_recorded = false;
        hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(1), 0, e->stream, e->fold ? e->ctl2 : e->ctl, c);
        e->fold_step = 0;
        if (G.comm) {                                              // the reduced statistics of the loaded state
            e->red_in = e->red_g;
            (void)hipMemcpyAsync(e->red_g, e->red, (size_t)e->red_count * sizeof(double), hipMemcpyDeviceToDevice, e->stream);
        }
        volatile double *ho = e->h_out;
        ho[5] = 0.0; ho[6] = 0.0; ho[7] = 0.0;
        e->run_active = true;
    }
    auto cleanup = [&](int rc) {
        if (e0->poisoned) {                                        // timed out: the streams may never drain -- leave at once
            std::string msg = last_error_cstr();
            for (int p = 0; p < G.count; p++) G.e[p]->poisoned = true;
            return fail(rc, "%s", msg.c_str());
        }
        for (int p = 0; p < G.count; p++) {
            vbnmf_engine *e = G.e[p];
            (void)hipStreamSynchronize(e->stream);
            if (e->cstream) (void)hipStreamSynchronize(e->cstream);
            if (G.comm) {                                          // host-stepped calls read the reduced statistics from `red`
                (void)hipMemcpy(e->red, e->red_g, (size_t)e->red_count * sizeof(double), hipMemcpyDeviceToDevice);
                e->red_in = nullptr;
            }
            e->run_active = false; e->timing = timing[p];
            e->stop_ptr = nullptr;
            e->seq = 0.0; e->h_out[7] = 0.0;                       // the step path's sequence flag restarts
        }
        return rc;
    };
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return cleanup(fail(VBNMF_ERR_HIP, "loading the loop control block failed: %s", hipGetErrorString(he)));

    int rc = drive_loop(e0, max_it, e0->fold, [&]() { return queue_vb_step(G, fudge, history != nullptr, max_it); });
    if (rc) return cleanup(rc);
    for (int p = 0; p < G.count; p++) {
        he = hipStreamSynchronize(G.e[p]->stream);
        if (he == hipSuccess && G.e[p]->cstream) he = hipStreamSynchronize(G.e[p]->cstream);
        if (he != hipSuccess) return cleanup(fail(VBNMF_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(he)));
    }
    const double *ho = e0->h_out;                                  // written by the last live k_control
    const int it = (int)ho[5];
    for (int q = 0; q < 4; q++) hyper[q] = ho[8 + q];
    if (it_out) *it_out = it;
    if (lk0_out) *lk0_out = ho[12];
    if (lkh_out) *lkh_out = ho[0];
    if (reason_out) *reason_out = (int)ho[6];
    if (history && it > 0) std::memcpy(history, e0->h_hist, (size_t)it * 9 * sizeof(double));
    return cleanup(VBNMF_OK);
}

}  // namespace

extern "C" {

// Device-driven form of the per-rank loop of vb_iterate (reference R/bayesian.R:336-352).
int vbnmf_engine_run(vbnmf_engine *e, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0, int32_t dn,
                     const int32_t *flags, int32_t *it_out, double *lk0_out, double *lkh_out, int32_t *reason_out,
                     double *history, int64_t history_rows)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    LoopGroup G{&e, 1, nullptr};
    if (e->partitioned) {
        if (!e->comm || e->comm->kind != 0)
            return fail(VBNMF_ERR_STATE, "the device-driven loop of a partitioned engine needs an RCCL communicator (vbnmf_engine_attach_comm), or vbnmf_group_run for a local group");
        if (int rc = use_device(e)) return rc;
        if (int rc = ensure_comm_resources(e)) return rc;
        G.comm = e->comm;
    }
    return run_group(G, hyper, fudge, max_it, tol, n0, dn, flags, it_out, lk0_out, lkh_out, reason_out, history, history_rows);
}

}  // extern "C"

// ---------------------------------------------------------------- a batch of engines stepped by one launch (kernels.h: k_update2_batch)
namespace {

#ifdef VBNMF_DEV_FEW_RANKS
#define VBNMF_FOR_EACH_R_BATCH(X) X(4) X(6) X(8) X(10)
#else
#define VBNMF_FOR_EACH_R_BATCH(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
#endif
constexpr int kBatchMaxPaddedRank = 16;      // the small-matrix regime the batch is for (instantiations cost build time)
constexpr int kBatchMax = 64;

template <int R, bool WIDE>
int launch_sweep_batch_t(vbnmf_engine *e, const SweepSide *jobs, int B)
{
    constexpr int NT = sweep_threads(R);
    static std::atomic<bool> attr_set[16];
    const void *fn = (const void *)k_sweep_batch<R, WIDE, NT, 1>;
    if (e->device >= 16 || !attr_set[e->device].load(std::memory_order_acquire)) {
        if (int rc = prepare_sweep_kernel(fn)) return rc;
        if (e->device < 16) attr_set[e->device].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_sweep_batch<R, WIDE, NT, 1>), dim3((unsigned)e->n_wg, (unsigned)B), dim3(NT), e->lds_bytes, e->stream, jobs);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_sweep_batch(vbnmf_engine *e, const SweepSide *jobs, int B)
{
    switch (e->R) {
#define X(RR) case RR: return e->wide ? launch_sweep_batch_t<RR, true>(e, jobs, B) : launch_sweep_batch_t<RR, false>(e, jobs, B);
        VBNMF_FOR_EACH_R_BATCH(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "a batch serves padded ranks up to %d (this engine: %d)", kBatchMaxPaddedRank, e->R);
    }
}

int launch_update2_batch(vbnmf_engine *e, const Upd2Job *jobs, int B)
{
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_update2_batch<RR>), dim3((unsigned)e->ub, (unsigned)B), dim3(kUpdateThreads), 0, e->stream, jobs); break;
        VBNMF_FOR_EACH_R_BATCH(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "a batch serves padded ranks up to %d (this engine: %d)", kBatchMaxPaddedRank, e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// Host side of a batch's device-driven loops (drive_loop's rule over all engines): steps queued in batches of eight, two batches
// ahead of the device; batch b + 2 is queued when some engine has neither stopped nor fallen short of the last step of batch b.
// `queue_batch` queues up to eight more steps and counts them in `queued`.  Every engine's result block is polled: [5] = steps done,
// [6] = reason (raised after [5]), [7] = steps done (raised last).  Returns with e0->poisoned set when the wait timed out.
template <class QueueBatch>
int drive_batch_loop(vbnmf_engine **engs, int B, hipStream_t S, int max_it, const int &queued, QueueBatch &&queue_batch, const char *what)
{
    vbnmf_engine *e0 = engs[0];
    if (int rc = queue_batch()) return rc;
    if (int rc = queue_batch()) return rc;
    const double limit = wait_timeout_s();
    for (int bt = 0;; bt++) {
        const int target = (int)std::min<int64_t>((int64_t)(bt + 1) * 8, max_it);
        const auto t0 = std::chrono::steady_clock::now();
        bool all_stopped = false;
        for (long spins = 1;; spins++) {
            bool reached = true;
            all_stopped = true;
            for (int b = 0; b < B; b++) {
                volatile double *ho = engs[b]->h_out;
                const bool stopped = ho[6] != 0.0;
                all_stopped = all_stopped && stopped;
                reached = reached && (stopped || (int)ho[7] >= target);
            }
            if (reached) break;
            if ((spins & 0xFFFF) == 0) {
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited > limit) {
                    e0->poisoned = true;
                    return fail(VBNMF_ERR_HIP, "timed out after %.1f s (VBNMF_WAIT_TIMEOUT_S) waiting for step %d of %s (%d queued)", waited, target, what, queued);
                }
                hipError_t q = hipStreamQuery(S);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(VBNMF_ERR_HIP, "%s failed on the device: %s", what, hipGetErrorString(q));
                if (q == hipSuccess) {                            // idle: everything queued has run; one more look, then a step is lost
                    bool ok = true;
                    const int expect = queued - (queued < max_it ? 1 : 0);      // (step t is reported by step t + 1's launch, the last by the closing ones)
                    for (int b = 0; b < B; b++) { volatile double *ho = engs[b]->h_out; ok = ok && (ho[6] != 0.0 || (int)ho[7] >= std::min(target, expect)); }
                    if (!ok) return fail(VBNMF_ERR_HIP, "the device went idle before the queued steps of %s finished", what);
                }
            }
        }
        if (all_stopped || target >= max_it) return VBNMF_OK;
        if (int rc = queue_batch()) return rc;
    }
}

// the fold of step t (1-based) of engine e's run, as queue_vb_step builds it
ControlFold batch_fold(const vbnmf_engine *e, int t, bool hist)
{
    ControlFold f{};
    f.prev = e->ctl2 + ((t - 1) & 1); f.next = e->ctl2 + (t & 1);
    f.epart = e->epart; f.nepart = 2 * (int64_t)e->n_wg;
    f.lgx = e->lgx; f.n = (double)e->n; f.m_global = (double)e->m_global;
    f.history = hist ? e->h_hist_dev : nullptr; f.out_host = e->h_out_dev;
    f.do_control = t > 1 ? 1 : 0;
    return f;
}

}  // namespace

extern "C" {

// Grids of the engines the CALLING host thread creates from now on: `n_wg` persistent workgroups of the sweep (the layouts
// are cut for that many) and `update_blocks` blocks of the update; 0 = the default (one per CU).  For the engines of a batch:
// B engines step in one launch of B x grid workgroups, so a batch of 8 wants grids of 32 -- a single small engine is
// faster on 256 (every block's share of a 15 us kernel is its prologue), eight of them are not (eight such blocks per CU).
// The grid fixes the order of the block-wise sums: engines of different grids agree to rounding, not bit for bit.
int vbnmf_set_engine_grid(int32_t n_wg, int32_t update_blocks)
{
    if (n_wg < 0 || update_blocks < 0 || update_blocks > kUpdateBlocks) return fail(VBNMF_ERR_BAD_ARG, "grid sizes must lie in [0, %d]", kUpdateBlocks);
    tl_grid_nwg = n_wg; tl_grid_ub = update_blocks;
    return VBNMF_OK;
}

// Row width of the engines the CALLING host thread creates from now on: their factors are stored `padded_rank` columns wide
// (a padded rank the kernels are built for, >= the engine's own; 0: back to the rank's own width), the columns beyond the rank
// held at zero as the padding always is.  Engines of DIFFERENT ranks made this way share kernels, layouts and update table, i.e.
// a batch: a whole rank sweep of a small matrix steps in one launch.  The width fixes the update's thread mapping, hence the
// order of its block-wise sums: a padded engine agrees with the unpadded one to rounding, not bit for bit.
int vbnmf_set_engine_padding(int32_t padded)
{
    if (padded != 0 && (padded < 2 || padded > VBNMF_MAX_RANK || padded_rank(padded) != padded))
        return fail(VBNMF_ERR_BAD_ARG, "%d is not a padded rank (even up to 32, a multiple of 8 up to 64, of 16 up to %d)", padded, VBNMF_MAX_RANK);
    tl_pad_R = padded;
    return VBNMF_OK;
}

// The device-driven loops of `count` engines of ONE rank on ONE matrix (the restarts of a rank: same layouts, grids and update
// table; each engine its own state), stepped TOGETHER: two launches per step for the whole batch, every engine's blocks
// following its own control block (its own hyper-parameters, evidence, stop).  Results per engine are those of
// vbnmf_engine_run on it alone, bit for bit.  hyper: [count][4] in / out; it / lk0 / lkh / reason: [count]; history (or
// null): [count][history_rows][9].  Engines must be unpartitioned, of the one-launch update form, of padded rank <= 16.
int vbnmf_batch_run(vbnmf_engine **engs, int32_t count, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0,
                    int32_t dn, const int32_t *flags, int32_t *it_out, double *lk0_out, double *lkh_out, int32_t *reason_out,
                    double *history, int64_t history_rows)
{
    if (!engs || !hyper || !flags) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (count < 1 || count > kBatchMax) return fail(VBNMF_ERR_BAD_ARG, "a batch holds 1 to %d engines", kBatchMax);
    if (max_it < 1 || dn < 1) return fail(VBNMF_ERR_BAD_ARG, "max_it and dn must be >= 1");
    if (history && history_rows < max_it) return fail(VBNMF_ERR_BAD_ARG, "history needs max_it rows of 9 doubles per engine");
    vbnmf_engine *e0 = engs[0];
    if (!e0) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    const int B = count;
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
        for (int q = 0; q < b; q++) if (engs[q] == e) return fail(VBNMF_ERR_BAD_ARG, "the same engine twice in a batch");
        if (int rc = use_device(e)) return rc;
        if (e->partitioned || e->comm || !e->pair || !e->fold || e->R > kBatchMaxPaddedRank)
            return fail(VBNMF_ERR_STATE, "a batch takes unpartitioned engines of the one-launch update form and padded rank <= %d", kBatchMaxPaddedRank);
        if (e->device != e0->device || e->n != e0->n || e->m != e0->m || e->R != e0->R || e->n_wg != e0->n_wg || e->ub != e0->ub ||
            e->NT != e0->NT || e->wide != e0->wide || e->lds_bytes != e0->lds_bytes || e->upd_stride4 != e0->upd_stride4 || e->upd_V != e0->upd_V ||
            e->upd_ids_off != e0->upd_ids_off || e->A.n_slices != e0->A.n_slices || e->B.n_slices != e0->B.n_slices)
            return fail(VBNMF_ERR_STATE, "the engines of a batch must be of one row width on one matrix (same padded rank, layouts, grids and update table)");
        if (!e->has_state || !e->stats_ready) return fail(VBNMF_ERR_STATE, "batch run before set_state");
        if (e->step_pending) return fail(VBNMF_ERR_STATE, "batch run between step_local and step_finish");
        if (history) { if (int rc = ensure_history(e, (size_t)max_it * 9)) return rc; }
    }
    hipStream_t S = e0->stream;
    for (int b = 1; b < B; b++) HIPCHECK(hipStreamSynchronize(engs[b]->stream));      // (idle already: set_state ends with a synchronise)

    // ---- the jobs: the update's of step 1 (nothing to evaluate yet), of the odd and of the even steps; the sweep's by parity
    const bool hist = history != nullptr;
    std::vector<Upd2Job> ju((size_t)3 * B);
    std::vector<SweepSide> js((size_t)2 * 2 * B);
    std::vector<hipStream_t> own(B);
    std::vector<bool> timing(B);
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        own[b] = e->stream; timing[b] = e->timing;
        e->timing = false; e->ev_recorded = false; e->ev2_recorded = false;
        double *Wt[2] = {e->bpW, e->bpW_alt}, *Ht[2] = {e->bpH, e->bpH_alt};          // [0]: the latest tables as the run starts
        for (int v = 0; v < 3; v++) {
            const int t = v == 0 ? 1 : (v == 1 ? 3 : 2);
            Upd2Job &J = ju[(size_t)v * B + b];
            J.W.part = e->A.part; J.W.nmaj = e->n; J.W.l = e->lw; J.W.ll = e->llw; J.W.e = e->ew; J.W.d = e->dw;
            J.H.part = e->B.part; J.H.nmaj = e->m; J.H.l = e->lh; J.H.ll = e->llh; J.H.e = e->eh; J.H.d = e->dh;
            J.W.bp_prev = Wt[(t - 1) & 1]; J.W.bp = Wt[t & 1];
            J.H.bp_prev = Ht[(t - 1) & 1]; J.H.bp = Ht[t & 1];
            J.T = UpdTable{e->upd_tab, e->upd_stride4, e->upd_V, e->upd_ids_off};
            J.r = e->r; J.nb = e->ub; J.ncs = e->n_wg; J.csum = e->csum; J.fudge = fudge;
            J.fold = batch_fold(e, t, hist);
        }
        e->run_active = true;
        for (int par = 0; par < 2; par++) {                       // the sweep of step t reads the stop flag that step's update left
            e->stop_ptr = &(e->ctl2 + par)->stop;
            js[((size_t)par * B + b) * 2] = sweep_side_args(e, e->A, true, e->epart);
            js[((size_t)par * B + b) * 2 + 1] = sweep_side_args(e, e->B, false, e->epart + e->n_wg);
        }
        e->stop_ptr = nullptr;
    }
    Upd2Job *d_ju = nullptr;
    SweepSide *d_js = nullptr;
    auto restore = [&](int rc) {
        for (int b = 0; b < B; b++) {
            vbnmf_engine *e = engs[b];
            e->stream = own[b];
            e->run_active = false; e->timing = timing[b]; e->stop_ptr = nullptr;
            e->seq = 0.0; e->h_out[7] = 0.0;
        }
        dev_free(d_ju); dev_free(d_js);
        return rc;
    };
    if (int rc = dev_upload(&d_ju, ju)) return restore(rc);
    if (int rc = dev_upload(&d_js, js)) return restore(rc);
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        LoopCtl c{};
        for (int q = 0; q < 4; q++) { c.hyper[q] = hyper[(size_t)b * 4 + q]; c.flags[q] = flags[q] ? 1 : 0; }
        c.lk0 = 0.0;
        c.tol = tol; c.max_it = max_it; c.n0 = n0; c.dn = dn;
        hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(1), 0, S, e->ctl2, c);
        e->fold_step = 0;
        volatile double *ho = e->h_out;
        ho[5] = 0.0; ho[6] = 0.0; ho[7] = 0.0;
        e->stream = S;                                           // every launch of the run goes on the first engine's stream
    }
    {
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) { (void)hipStreamSynchronize(S); return restore(fail(VBNMF_ERR_HIP, "loading the loop control blocks failed: %s", hipGetErrorString(he))); }
    }

    // ---- the loop: steps queued in batches of eight, two batches ahead of the device (drive_loop's rule, over all engines)
    int queued = 0;
    auto queue_step = [&]() -> int {
        const int t = queued + 1;
        const int v = t == 1 ? 0 : ((t & 1) ? 1 : 2);
        if (int rc = launch_update2_batch(e0, d_ju + (size_t)v * B, B)) return rc;
        if (int rc = launch_sweep_batch(e0, d_js + (size_t)(t & 1) * B * 2, B)) return rc;
        for (int b = 0; b < B; b++) {
            vbnmf_engine *e = engs[b];
            std::swap(e->bpW, e->bpW_alt); std::swap(e->bpH, e->bpH_alt);      // (e->bpW / e->bpH name the latest tables, as launch_update2 keeps them)
            e->fold_step = t;
        }
        if (t == max_it) {                                       // behind the last step: every engine's control step alone
            for (int b = 0; b < B; b++) {
                vbnmf_engine *e = engs[b];
                ControlFold g = batch_fold(e, t + 1, hist);
                g.bpW_prev = e->bpW;
                g.do_control = 1; g.control_only = 1;
                if (int rc = launch_update(e, true, 0, 0, fudge, nullptr, &g)) return rc;
            }
        }
        queued++;
        return VBNMF_OK;
    };
    auto queue_batch = [&]() -> int {
        for (int q = 0; q < 8 && queued < max_it; q++) if (int rc = queue_step()) return rc;
        return VBNMF_OK;
    };
    auto fail_out = [&](int rc) {
        std::string msg = last_error_cstr();
        if (e0->poisoned) { for (int b = 0; b < B; b++) { engs[b]->poisoned = true; engs[b]->stream = own[b]; } return fail(rc, "%s", msg.c_str()); }
        (void)hipStreamSynchronize(S);
        restore(rc);
        return fail(rc, "%s", msg.c_str());
    };
    if (int rc = drive_batch_loop(engs, B, S, max_it, queued, queue_batch, "a batch's device-driven loops")) return fail_out(rc);
    {
        hipError_t he = hipStreamSynchronize(S);
        if (he != hipSuccess) return restore(fail(VBNMF_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(he)));
    }
    for (int b = 0; b < B; b++) {
        const double *ho = engs[b]->h_out;
        const int it = (int)ho[5];
        for (int q = 0; q < 4; q++) hyper[(size_t)b * 4 + q] = ho[8 + q];
        if (it_out) it_out[b] = it;
        if (lk0_out) lk0_out[b] = ho[12];
        if (lkh_out) lkh_out[b] = ho[0];
        if (reason_out) reason_out[b] = (int)ho[6];
        if (history && it > 0) std::memcpy(history + (size_t)b * (size_t)history_rows * 9, engs[b]->h_hist, (size_t)it * 9 * sizeof(double));
    }
    return restore(VBNMF_OK);
}

}  // extern "C"

namespace {

template <int R, bool WIDE, bool LOGTERM>
int launch_sweep1_batch_t(vbnmf_engine *e, const SweepSide *jobs, int B)
{
    constexpr int NT = sweep_threads(R);
    static std::atomic<bool> attr_set[16];
    const void *fn = (const void *)k_sweep1_batch<R, WIDE, LOGTERM, NT>;
    if (e->device >= 16 || !attr_set[e->device].load(std::memory_order_acquire)) {
        if (int rc = prepare_sweep_kernel(fn)) return rc;
        if (e->device < 16) attr_set[e->device].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_sweep1_batch<R, WIDE, LOGTERM, NT>), dim3((unsigned)e->n_wg, (unsigned)B), dim3(NT), e->lds_bytes, e->stream, jobs);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_sweep1_batch(vbnmf_engine *e, const SweepSide *jobs, int B, bool logterm)
{
    switch (e->R) {
#define X(RR) case RR: return e->wide ? (logterm ? launch_sweep1_batch_t<RR, true, true>(e, jobs, B) : launch_sweep1_batch_t<RR, true, false>(e, jobs, B)) \
                                      : (logterm ? launch_sweep1_batch_t<RR, false, true>(e, jobs, B) : launch_sweep1_batch_t<RR, false, false>(e, jobs, B));
        VBNMF_FOR_EACH_R_BATCH(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "a batch serves padded ranks up to %d (this engine: %d)", kBatchMaxPaddedRank, e->R);
    }
}

int launch_ml_update_batch(vbnmf_engine *e, const MlUpdJob *jobs, int B)
{
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_ml_update_batch<RR>), dim3((unsigned)e->ub, (unsigned)B), dim3(kUpdateThreads), 0, e->stream, jobs); break;
        VBNMF_FOR_EACH_R_BATCH(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "a batch serves padded ranks up to %d (this engine: %d)", kBatchMaxPaddedRank, e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

MlFold batch_ml_fold(const vbnmf_engine *e, int t, bool hist, const double *bpH_prev)
{
    MlFold f{};
    f.prev = e->ctl2 + ((t - 1) & 1); f.next = e->ctl2 + (t & 1);
    f.bpH_prev = bpH_prev;
    f.epart = e->epart + e->n_wg; f.nepart = (int64_t)e->n_wg;
    f.xlx = e->xlx; f.n = (double)e->n; f.m = (double)e->m;
    f.history = hist ? e->h_hist_dev : nullptr; f.out_host = e->h_out_dev;
    f.do_control = t > 1 ? 1 : 0;
    return f;
}

}  // namespace

extern "C" {

// The device-driven ML-NMF loops (vbnmf_engine_ml_run; factorize() under criterion = 'likelihood', reference
// R/factorize.R:194-213) of `count` engines of ONE rank on ONE matrix -- the `nrun` restarts factorize() makes of every rank
// (R/factorize.R:181; its default is nrun = 20) -- stepped together: four launches per step for the whole batch.  Per engine
// the results are those of vbnmf_engine_ml_run on it alone, bit for bit.  it_out, lk_out, reason_out: [count]; history (or
// NULL): [count][history_rows], history_rows >= max_it.  Engines as for vbnmf_batch_run, their states set by ml_set_state.
int vbnmf_batch_ml_run(vbnmf_engine **engs, int32_t count, int32_t prior, double gamma_a, double gamma_b, int32_t max_it, double tol,
                       int32_t *it_out, double *lk_out, int32_t *reason_out, double *history, int64_t history_rows)
{
    if (!engs) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (count < 1 || count > kBatchMax) return fail(VBNMF_ERR_BAD_ARG, "a batch holds 1 to %d engines", kBatchMax);
    if (max_it < 1) return fail(VBNMF_ERR_BAD_ARG, "max_it must be >= 1");
    if (history && history_rows < max_it) return fail(VBNMF_ERR_BAD_ARG, "history needs max_it doubles per engine");
    vbnmf_engine *e0 = engs[0];
    if (!e0) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    const int B = count;
    const double eps = 2.220446049250313e-16;
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
        for (int q = 0; q < b; q++) if (engs[q] == e) return fail(VBNMF_ERR_BAD_ARG, "the same engine twice in a batch");
        if (int rc = use_device(e)) return rc;
        if (e->partitioned || e->comm || !e->fold || !e->bpH_alt || e->R > kBatchMaxPaddedRank)
            return fail(VBNMF_ERR_STATE, "a batch takes unpartitioned engines with the control step folded in and padded rank <= %d", kBatchMaxPaddedRank);
        if (e->device != e0->device || e->n != e0->n || e->m != e0->m || e->R != e0->R || e->n_wg != e0->n_wg || e->ub != e0->ub ||
            e->NT != e0->NT || e->wide != e0->wide || e->lds_bytes != e0->lds_bytes || e->A.n_slices != e0->A.n_slices || e->B.n_slices != e0->B.n_slices)
            return fail(VBNMF_ERR_STATE, "the engines of a batch must be of one row width on one matrix (same padded rank, layouts and grids)");
        if (!e->ml_ready) return fail(VBNMF_ERR_STATE, "batch ML run before ml_set_state");
        if (history) { if (int rc = ensure_history(e, (size_t)max_it)) return rc; }
    }
    hipStream_t S = e0->stream;
    for (int b = 1; b < B; b++) HIPCHECK(hipStreamSynchronize(engs[b]->stream));

    const bool hist = history != nullptr;
    static const int stage_allowed = [] { const char *v = getenv("VBNMF_NO_STAGE_IDS"); return (v && v[0] == '1') ? 0 : 1; }();
    // jobs: the H update's of step 1, of the odd and of the even steps; the W update's and the two sweeps' by parity
    std::vector<MlUpdJob> jh((size_t)3 * B), jw((size_t)2 * B);
    std::vector<SweepSide> jg((size_t)2 * B), jc((size_t)2 * B);
    std::vector<hipStream_t> own(B);
    std::vector<bool> timing(B);
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        own[b] = e->stream; timing[b] = e->timing;
        e->timing = false; e->ev_recorded = false; e->ev2_recorded = false;
        double *Ht[2] = {e->bpH, e->bpH_alt};                                    // [0]: the latest table as the run starts
        for (int v = 0; v < 3; v++) {
            const int t = v == 0 ? 1 : (v == 1 ? 3 : 2);
            MlUpdJob &J = jh[(size_t)v * B + b];
            J.part = e->B.part; J.inv_ptr = e->B.inv_ptr; J.inv_task = e->B.inv_task; J.nmaj = e->m;
            J.other_bp = e->bpW; J.f = e->lh; J.bp = Ht[t & 1]; J.stop = nullptr;
            J.ga = gamma_a; J.gb = gamma_b; J.eps = eps;
            J.r = e->r; J.other_nb = e->ub; J.prior = prior;
            J.stage_ids = stage_allowed && e->B.n_tasks >= (int64_t)256 * e->ub ? 1 : 0;
            J.fold = batch_ml_fold(e, t, hist, Ht[(t - 1) & 1]);
        }
        e->run_active = true;
        for (int par = 0; par < 2; par++) {                                      // step t of parity par = t & 1
            const int32_t *stop = &(e->ctl2 + par)->stop;                        // what that step's H update left
            MlUpdJob &J = jw[(size_t)par * B + b];
            J.part = e->A.part; J.inv_ptr = e->A.inv_ptr; J.inv_task = e->A.inv_task; J.nmaj = e->n;
            J.other_bp = Ht[par]; J.f = e->lw; J.bp = e->bpW; J.stop = stop;
            J.ga = gamma_a; J.gb = gamma_b; J.eps = eps;
            J.r = e->r; J.other_nb = e->ub; J.prior = prior;
            J.stage_ids = stage_allowed && e->A.n_tasks >= (int64_t)256 * e->ub ? 1 : 0;
            J.fold = MlFold{};
            e->stop_ptr = stop;
            SweepSide g = sweep_side_args(e, e->A, true, e->epart);
            g.logterm = 0;
            SweepSide c = sweep_side_args(e, e->B, false, e->epart + e->n_wg);
            c.logterm = 1;
            jg[(size_t)par * B + b] = g; jc[(size_t)par * B + b] = c;
        }
        e->stop_ptr = nullptr;
    }
    MlUpdJob *d_jh = nullptr, *d_jw = nullptr;
    SweepSide *d_jg = nullptr, *d_jc = nullptr;
    auto restore = [&](int rc) {
        for (int b = 0; b < B; b++) {
            vbnmf_engine *e = engs[b];
            e->stream = own[b];
            e->run_active = false; e->timing = timing[b]; e->stop_ptr = nullptr;
            e->seq = 0.0; e->h_out[7] = 0.0;
        }
        dev_free(d_jh); dev_free(d_jw); dev_free(d_jg); dev_free(d_jc);
        return rc;
    };
    if (int rc = dev_upload(&d_jh, jh)) return restore(rc);
    if (int rc = dev_upload(&d_jw, jw)) return restore(rc);
    if (int rc = dev_upload(&d_jg, jg)) return restore(rc);
    if (int rc = dev_upload(&d_jc, jc)) return restore(rc);
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        LoopCtl c{};
        c.lk0 = -INFINITY;                                                        // lkold <- -Inf (R/factorize.R:193)
        c.tol = tol; c.max_it = max_it;
        hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(1), 0, S, e->ctl2, c);
        e->fold_step = 0;
        volatile double *ho = e->h_out;
        ho[5] = 0.0; ho[6] = 0.0; ho[7] = 0.0;
        e->stream = S;
    }
    {
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) { (void)hipStreamSynchronize(S); return restore(fail(VBNMF_ERR_HIP, "loading the loop control blocks failed: %s", hipGetErrorString(he))); }
    }
    int queued = 0;
    auto queue_step = [&]() -> int {
        const int t = queued + 1;
        const int v = t == 1 ? 0 : ((t & 1) ? 1 : 2), par = t & 1;
        if (int rc = launch_ml_update_batch(e0, d_jh + (size_t)v * B, B)) return rc;           // H <- , the previous step's control folded in
        for (int b = 0; b < B; b++) { vbnmf_engine *e = engs[b]; std::swap(e->bpH, e->bpH_alt); e->fold_step = t; }
        if (int rc = launch_sweep1_batch(e0, d_jg + (size_t)par * B, B, false)) return rc;     // gene side on (w, h_new)
        if (int rc = launch_ml_update_batch(e0, d_jw + (size_t)par * B, B)) return rc;         // W <-
        if (int rc = launch_sweep1_batch(e0, d_jc + (size_t)par * B, B, true)) return rc;      // cell side on (h_new, w_new): next step's statistics + sum x log(wh)
        if (t == max_it) {
            for (int b = 0; b < B; b++) {
                vbnmf_engine *e = engs[b];
                MlFold g = batch_ml_fold(e, t + 1, hist, e->bpH);
                g.do_control = 1; g.control_only = 1;
                if (int rc = launch_ml_update(e, false, prior, gamma_a, gamma_b, eps, &g)) return rc;
            }
        }
        queued++;
        return VBNMF_OK;
    };
    auto queue_batch = [&]() -> int {
        for (int q = 0; q < 8 && queued < max_it; q++) if (int rc = queue_step()) return rc;
        return VBNMF_OK;
    };
    auto fail_out = [&](int rc) {
        std::string msg = last_error_cstr();
        if (e0->poisoned) { for (int b = 0; b < B; b++) { engs[b]->poisoned = true; engs[b]->stream = own[b]; } return fail(rc, "%s", msg.c_str()); }
        (void)hipStreamSynchronize(S);
        restore(rc);
        return fail(rc, "%s", msg.c_str());
    };
    if (int rc = drive_batch_loop(engs, B, S, max_it, queued, queue_batch, "a batch's ML loops")) return fail_out(rc);
    {
        hipError_t he = hipStreamSynchronize(S);
        if (he != hipSuccess) return restore(fail(VBNMF_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(he)));
    }
    for (int b = 0; b < B; b++) {
        vbnmf_engine *e = engs[b];
        const int it = (int)e->h_out[5];
        if (it_out) it_out[b] = it;
        if (lk_out) lk_out[b] = e->h_out[0];
        if (reason_out) reason_out[b] = (int)e->h_out[6];
        if (history && it > 0) std::memcpy(history + (size_t)b * (size_t)history_rows, e->h_hist, (size_t)it * sizeof(double));
    }
    std::vector<double> lk_last(B);
    for (int b = 0; b < B; b++) lk_last[b] = engs[b]->h_out[0];
    const int rc = restore(VBNMF_OK);
    for (int b = 0; b < B; b++) engs[b]->h_out[0] = lk_last[b];                   // ml_likelihood() keeps answering for the pair held now
    return rc;
}

// ---------------------------------------------------------------- communicators (comm.h)
int vbnmf_comm_unique_id(void *id, int64_t bytes)
{
    if (!id || bytes < (int64_t)sizeof(ncclUniqueId)) return fail(VBNMF_ERR_BAD_ARG, "the id buffer needs %d bytes", (int)sizeof(ncclUniqueId));
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(VBNMF_ERR_NO_DEVICE, "%s", api.error.c_str());
    ncclUniqueId u;
    if (int rc = rccl_check(api.GetUniqueId(&u), "ncclGetUniqueId")) return rc;
    std::memcpy(id, &u, sizeof u);
    return VBNMF_OK;
}

int vbnmf_comm_create(const void *id, int64_t bytes, int32_t nranks, int32_t rank, int32_t device, vbnmf_comm **out)
{
    if (!out) return fail(VBNMF_ERR_BAD_ARG, "out pointer is NULL");
    *out = nullptr;
    if (!id || bytes < (int64_t)sizeof(ncclUniqueId)) return fail(VBNMF_ERR_BAD_ARG, "the id buffer needs %d bytes", (int)sizeof(ncclUniqueId));
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [0, %d)", rank, nranks);
    if (int rc = check_device(device)) return rc;
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(VBNMF_ERR_NO_DEVICE, "%s", api.error.c_str());
    HIPCHECK(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    vbnmf_comm *c = new (std::nothrow) vbnmf_comm();
    if (!c) return fail(VBNMF_ERR_OOM, "out of host memory");
    c->kind = 0; c->nranks = nranks; c->rank = rank; c->device = device;
    if (int rc = rccl_check(api.CommInitRank(&c->nc, nranks, u, rank), "ncclCommInitRank")) { delete c; return rc; }
    *out = c;
    return VBNMF_OK;
}

int vbnmf_comm_create_local(int32_t nranks, int32_t device, vbnmf_comm **out)
{
    if (!out) return fail(VBNMF_ERR_BAD_ARG, "out pointer is NULL");
    *out = nullptr;
    if (nranks < 1 || nranks > 64) return fail(VBNMF_ERR_BAD_ARG, "a local group holds 1..64 partitions");
    if (int rc = check_device(device)) return rc;
    vbnmf_comm *c = new (std::nothrow) vbnmf_comm();
    if (!c) return fail(VBNMF_ERR_OOM, "out of host memory");
    c->kind = 1; c->nranks = nranks; c->rank = 0; c->device = device;
    *out = c;
    return VBNMF_OK;
}

void vbnmf_comm_destroy(vbnmf_comm *c)
{
    if (!c) return;
    for (vbnmf_engine *e : c->members) if (e && e->comm == c) e->comm = nullptr;
    if (c->kind == 0 && c->nc) (void)rccl_api().CommDestroy(c->nc);
    dev_free(c->d_send_big); dev_free(c->d_send_small); dev_free(c->d_recv_big); dev_free(c->d_recv_small);
    delete c;
}

int vbnmf_comm_info(const vbnmf_comm *c, int32_t *nranks, int32_t *rank, int32_t *kind)
{
    if (!c) return fail(VBNMF_ERR_BAD_ARG, "communicator handle is NULL");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (kind) *kind = c->kind;
    return VBNMF_OK;
}

int vbnmf_engine_attach_comm(vbnmf_engine *e, vbnmf_comm *c)
{
    if (!e || !c) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (e->comm) return fail(VBNMF_ERR_STATE, "the engine already has a communicator");
    {   // engines that were destroyed while attached leave empty seats: a communicator none of whose engines is alive starts over
        bool any = false;
        for (vbnmf_engine *q : c->members) any |= q != nullptr;
        if (!any) c->members.clear();
    }
    if (e->device != c->device) return fail(VBNMF_ERR_BAD_ARG, "engine on device %d, communicator on device %d", e->device, c->device);
    if (c->kind == 0 && !c->members.empty()) return fail(VBNMF_ERR_STATE, "an RCCL communicator serves one engine per process");
    if (c->kind == 1 && (int)c->members.size() >= c->nranks) return fail(VBNMF_ERR_STATE, "the local group is full");
    if (c->kind == 1 && !c->members.empty() && (c->members[0]->n != e->n || c->members[0]->R != e->R || c->members[0]->m_global != e->m_global))
        return fail(VBNMF_ERR_BAD_ARG, "partition engines of one group must share genes, rank and the global cell count");
    if (int rc = use_device(e)) return rc;
    if (int rc = ensure_comm_resources(e)) return rc;
    e->comm = c;
    e->comm_rank = c->kind == 0 ? c->rank : (int)c->members.size();
    c->members.push_back(e);
    c->tables_ready = false;
    return VBNMF_OK;
}

// In-place all-reduce of the reduce buffer on the engine's stream: the exchange between step_local and step_finish
// (and between set_state and state_finish) of a host-stepped partitioned run, from C (RCCL communicators).
int vbnmf_engine_allreduce(vbnmf_engine *e)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->comm || e->comm->kind != 0) return fail(VBNMF_ERR_STATE, "allreduce needs an RCCL communicator attached to the engine");
    if (int rc = use_device(e)) return rc;
    return rccl_check(rccl_api().AllReduce(e->red, e->red, (size_t)e->red_count, ncclDouble, ncclSum, e->comm->nc, e->stream), "ncclAllReduce");
}

static int group_members(vbnmf_comm *c, LoopGroup &G)
{
    if (!c) return fail(VBNMF_ERR_BAD_ARG, "communicator handle is NULL");
    if (c->kind != 1) return fail(VBNMF_ERR_STATE, "not a local group (an RCCL communicator is driven through its engine)");
    if ((int)c->members.size() != c->nranks) return fail(VBNMF_ERR_STATE, "the local group has %d of its %d partitions attached", (int)c->members.size(), c->nranks);
    for (vbnmf_engine *q : c->members) if (!q) return fail(VBNMF_ERR_STATE, "a partition engine of the group has been destroyed");
    G.e = c->members.data(); G.count = c->nranks; G.comm = c;
    return VBNMF_OK;
}

// Local group: the state exchange after set_state on every member (sums the reduce buffers in place, partition
// order), then state_finish on each.
int vbnmf_group_state_finish(vbnmf_comm *c)
{
    LoopGroup G{};
    if (int rc = group_members(c, G)) return rc;
    if (int rc = use_device(G.e[0])) return rc;
    if (int rc = group_tables(c)) return rc;
    for (int p = 0; p < G.count; p++) {
        if (!G.e[p]->prime_pending) return fail(VBNMF_ERR_STATE, "group_state_finish: partition %d has no pending set_state", p);
        HIPCHECK(hipStreamSynchronize(G.e[p]->stream));
    }
    vbnmf_engine *L = G.e[0];
    // in place: every element is read from all partitions before it is written to any
    std::vector<double *> ptr(G.count);
    for (int p = 0; p < G.count; p++) ptr[p] = G.e[p]->red;
    double **d_ptr = nullptr;
    if (int rc = dev_alloc(&d_ptr, (size_t)G.count)) return rc;
    hipError_t he = hipMemcpy(d_ptr, ptr.data(), G.count * sizeof(void *), hipMemcpyHostToDevice);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(k_group_sum, dim3((unsigned)((L->red_count + 255) / 256)), dim3(256), 0, L->stream, (const double *const *)d_ptr, d_ptr, G.count, L->red_count);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(L->stream);
    dev_free(d_ptr);
    if (he != hipSuccess) return fail(VBNMF_ERR_HIP, "group state exchange failed: %s", hipGetErrorString(he));
    for (int p = 0; p < G.count; p++)
        if (int rc = vbnmf_engine_state_finish(G.e[p])) return rc;
    return VBNMF_OK;
}

int vbnmf_group_run(vbnmf_comm *c, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0, int32_t dn,
                    const int32_t *flags, int32_t *it_out, double *lk0_out, double *lkh_out, int32_t *reason_out,
                    double *history, int64_t history_rows)
{
    LoopGroup G{};
    if (int rc = group_members(c, G)) return rc;
    if (int rc = use_device(G.e[0])) return rc;
    if (int rc = group_tables(c)) return rc;
    return run_group(G, hyper, fudge, max_it, tol, n0, dn, flags, it_out, lk0_out, lkh_out, reason_out, history, history_rows);
}

int vbnmf_engine_get_state(vbnmf_engine *e, double *lw, double *lh, double *ew, double *eh, double *dw, double *dh)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->has_state) return fail(VBNMF_ERR_STATE, "get_state before set_state");
    if (int rc = use_device(e)) return rc;
    {
        struct Item { double *dst; const double *src; bool gene; };
        const Item items[6] = {{lw, e->lw, true}, {ew, e->ew, true}, {dw, e->dw, true},
                               {lh, e->lh, false}, {eh, e->eh, false}, {dh, e->dh, false}};
        size_t need = 0;
        for (const Item &it : items) if (it.dst) need += (size_t)(it.gene ? e->n : e->m) * e->R;
        if (int rc = ensure_stage(e, need)) return rc;
        size_t off = 0;
        for (const Item &it : items) {                   // every requested array into the pinned staging buffer, asynchronously ...
            if (!it.dst) continue;
            const size_t cnt = (size_t)(it.gene ? e->n : e->m) * e->R;
            HIPCHECK(hipMemcpyAsync(e->h_stage + off, it.src, cnt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            off += cnt;
        }
        HIPCHECK(hipStreamSynchronize(e->stream));        // ... one wait (it also covers whatever was queued before) ...
        off = 0;
        for (const Item &it : items) {                   // ... then into the caller's column-major arrays
            if (!it.dst) continue;
            const int64_t nmaj = it.gene ? e->n : e->m;
            from_index_major(e->h_stage + off, nmaj, e->r, e->R, !it.gene, it.dst, it.gene ? nullptr : &e->cell_perm);
            off += (size_t)nmaj * e->R;
        }
    }
    return VBNMF_OK;
}

int vbnmf_engine_timing_enable(vbnmf_engine *e, int32_t on)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    e->timing = on != 0;
    e->sweep_ms = 0.0; e->sweep_launches = 0; e->ev_recorded = false; e->ev2_recorded = false;
    return VBNMF_OK;
}

int vbnmf_engine_timing_get(vbnmf_engine *e, double *sweep_ms, int64_t *sweep_launches)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (sweep_ms) *sweep_ms = e->sweep_ms;
    if (sweep_launches) *sweep_launches = e->sweep_launches;
    e->sweep_ms = 0.0; e->sweep_launches = 0;
    return VBNMF_OK;
}

int vbnmf_engine_layout_info(const vbnmf_engine *e, int64_t *nnz, int64_t *slots_gene, int64_t *slots_cell,
                             int64_t *stream_bytes, int64_t *tiles_gene, int64_t *tiles_cell)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (nnz) *nnz = e->nnz;
    if (slots_gene) *slots_gene = e->A.n_slots;
    if (slots_cell) *slots_cell = e->B.n_slots;
    if (stream_bytes) *stream_bytes = (e->A.n_slots + e->B.n_slots) * (int64_t)(e->wide ? 12 : 4);
    if (tiles_gene) *tiles_gene = e->A.n_tasks;
    if (tiles_cell) *tiles_cell = e->B.n_tasks;
    return VBNMF_OK;
}

// Diagnostic: per-workgroup / per-wave 100 MHz timestamps of the last sweep (engine created with
// VBNMF_DEBUG_TIMES=1): out[side][wg][2 + 2*waves] = {wg start, wg end, (wave start, wave end)...}.
int vbnmf_engine_debug_times(vbnmf_engine *e, unsigned long long *out, int64_t capacity, int32_t *n_wg, int32_t *waves)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->dbg) return fail(VBNMF_ERR_STATE, "engine was not created with VBNMF_DEBUG_TIMES=1");
    if (n_wg) *n_wg = e->n_wg;
    if (waves) *waves = e->NT / 64;
    if (out) {
        if ((size_t)capacity < e->dbg_count) return fail(VBNMF_ERR_BAD_ARG, "buffer too small");
        if (int rc = use_device(e)) return rc;
        HIPCHECK(hipStreamSynchronize(e->stream));
        HIPCHECK(hipMemcpy(out, e->dbg, e->dbg_count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    return VBNMF_OK;
}

// ---------------------------------------------------------------- ML-NMF on the same engine
// (reference R/factorize.R:2-27 nmf_updateR, :40-49 likelihood; the factors live in the lw / lh arrays)
int vbnmf_engine_ml_set_state(vbnmf_engine *e, const double *w, const double *h)
{
    if (!e || !w || !h) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (e->partitioned) return fail(VBNMF_ERR_STATE, "ML-NMF needs an unpartitioned engine");
    if (int rc = use_device(e)) return rc;
    e->has_state = false; e->stats_ready = false; e->step_pending = false; e->prime_pending = false; e->ml_ready = false; e->ids_valid = false;
    try {
        std::vector<double> tmp;
        to_index_major(w, e->n, e->r, e->R, false, tmp);
        HIPCHECK(hipMemcpyAsync(e->lw, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
        to_index_major(h, e->m, e->r, e->R, true, tmp, &e->cell_perm);
        HIPCHECK(hipMemcpyAsync(e->lh, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory staging the state");
    }
    // colSums(w) block partials, then the cell-side statistics the first H update starts from
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_prime<RR>), dim3(e->ub), dim3(kUpdateThreads), 0, e->stream, e->n, e->r, e->lw, e->llw, e->lw, e->bpW); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    switch (e->R) {                                  // rowSums(h) block partials, for the likelihood of the loaded pair
#define X(RR) case RR: hipLaunchKernelGGL((k_prime<RR>), dim3(e->ub), dim3(kUpdateThreads), 0, e->stream, e->m, e->r, e->lh, e->llh, e->lh, e->bpH); break;
        VBNMF_FOR_EACH_R(X)
#undef X
    }
    HIPCHECK(hipGetLastError());
    const bool timing = e->timing;
    e->timing = false;                               // the priming sweep is not a step
    int rc = launch_sweep1(e, false);
    e->timing = timing;
    if (rc) return rc;
    if ((rc = launch_ml_final(e))) return rc;
    if ((rc = wait_result(e))) return rc;
    e->ml_ready = true;
    return VBNMF_OK;
}

int vbnmf_engine_ml_likelihood(vbnmf_engine *e, double *lk)
{
    if (!e || !lk) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (!e->ml_ready) return fail(VBNMF_ERR_STATE, "ml_likelihood before ml_set_state");
    *lk = e->h_out[0];                               // left there by the last k_ml_final (set_state or step)
    return VBNMF_OK;
}

int vbnmf_engine_ml_step(vbnmf_engine *e, int32_t prior, double gamma_a, double gamma_b, double *lk)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->ml_ready) return fail(VBNMF_ERR_STATE, "ml_step before ml_set_state");
    if (int rc = use_device(e)) return rc;
    const double eps = 2.220446049250313e-16;        // .Machine$double.eps (R/factorize.R:15,24)
    if (int rc = launch_ml_update(e, false, prior, gamma_a, gamma_b, eps)) return rc;   // H first (:8-15)
    if (int rc = launch_sweep1(e, true)) return rc;
    if (int rc = launch_ml_update(e, true, prior, gamma_a, gamma_b, eps)) return rc;    // then W on the new h (:17-24)
    if (int rc = launch_sweep1(e, false)) return rc;
    if (int rc = launch_ml_final(e)) return rc;
    if (int rc = wait_result(e)) return rc;
    if (int rc = harvest_timing(e)) return rc;
    if (lk) *lk = e->h_out[0];
    return VBNMF_OK;
}

// Device-driven form of factorize()'s inner loop under criterion = 'likelihood' (reference R/factorize.R:194-213).
int vbnmf_engine_ml_run(vbnmf_engine *e, int32_t prior, double gamma_a, double gamma_b, int32_t max_it, double tol,
                        int32_t *it_out, double *lk_out, int32_t *reason_out, double *history, int64_t history_rows)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->ml_ready) return fail(VBNMF_ERR_STATE, "ml_run before ml_set_state");
    if (max_it < 1) return fail(VBNMF_ERR_BAD_ARG, "max_it must be >= 1");
    if (history && history_rows < max_it) return fail(VBNMF_ERR_BAD_ARG, "history needs max_it doubles");
    if (int rc = use_device(e)) return rc;
    const double eps = 2.220446049250313e-16;

    LoopCtl c{};
    c.lk0 = -INFINITY;                                             // lkold <- -Inf (:193)
    c.tol = tol; c.max_it = max_it;
    if (history) { if (int rc = ensure_history(e, (size_t)max_it)) return rc; }
    hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(1), 0, e->stream, e->fold ? e->ctl2 : e->ctl, c);
    e->fold_step = 0;
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return fail(VBNMF_ERR_HIP, "loading the loop control block failed: %s", hipGetErrorString(he));
    volatile double *ho = e->h_out;
    ho[5] = 0.0; ho[6] = 0.0; ho[7] = 0.0;
    e->run_active = true;
    const bool timing = e->timing;
    e->timing = false;                                             // event pairs cannot follow launches queued ahead
    e->ev_recorded = false; e->ev2_recorded = false;
    auto done_with = [&](int rc) {
        if (e->poisoned) return rc;                                 // timed out: the stream may never drain
        (void)hipStreamSynchronize(e->stream);
        e->timing = timing; e->run_active = false;
        e->stop_ptr = nullptr;
        e->seq = 0.0; e->h_out[7] = 0.0;                            // the step path's sequence flag restarts
        return rc;
    };
    double *hist_dev = history ? e->h_hist_dev : nullptr;
    int rc = drive_loop(e, max_it, e->fold, [&]() -> int {
        if (e->fold) {
            // k_ml_update(H, with the control step of the PREVIOUS cell-side sweep folded in)  sweep  k_ml_update(W)  sweep ;
            // behind the last step of the run the control step alone (mlnmf.h: MlFold)
            const int t = ++e->fold_step;
            MlFold f{};
            f.prev = e->ctl2 + ((t - 1) & 1); f.next = e->ctl2 + (t & 1);
            f.bpH_prev = e->bpH;
            std::swap(e->bpH, e->bpH_alt);                          // this step's H-side partials go to the other table
            f.epart = e->epart + e->n_wg; f.nepart = (int64_t)e->n_wg;
            f.xlx = e->xlx; f.n = (double)e->n; f.m = (double)e->m;
            f.history = hist_dev; f.out_host = e->h_out_dev;
            f.do_control = t > 1 ? 1 : 0;
            int q = launch_ml_update(e, false, prior, gamma_a, gamma_b, eps, &f);
            e->stop_ptr = &f.next->stop;
            if (!q) q = launch_sweep1(e, true);
            if (!q) q = launch_ml_update(e, true, prior, gamma_a, gamma_b, eps);
            if (!q) q = launch_sweep1(e, false);
            if (!q && t == max_it) {
                MlFold g = f;
                g.prev = f.next; g.next = e->ctl2 + ((t + 1) & 1);
                g.bpH_prev = e->bpH;
                g.do_control = 1; g.control_only = 1;
                q = launch_ml_update(e, false, prior, gamma_a, gamma_b, eps, &g);
            }
            return q;
        }
        int q = launch_ml_update(e, false, prior, gamma_a, gamma_b, eps);
        if (!q) q = launch_sweep1(e, true);
        if (!q) q = launch_ml_update(e, true, prior, gamma_a, gamma_b, eps);
        if (!q) q = launch_sweep1(e, false);
        if (q) return q;
        switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_ml_control<RR>), dim3(1), dim3(1024), 0, e->stream, e->bpW, e->bpH, e->ub, e->epart + e->n_wg, (int64_t)e->n_wg, e->xlx, e->r, (double)e->n, (double)e->m, e->ctl, hist_dev, e->h_out_dev); break;
            VBNMF_FOR_EACH_R(X)
#undef X
            default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
        }
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(VBNMF_ERR_HIP, "k_ml_control launch failed: %s", hipGetErrorString(le));
        return VBNMF_OK;
    });
    if (rc) return done_with(rc);
    he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) return done_with(fail(VBNMF_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(he)));
    const int it = (int)e->h_out[5];
    const double lk_last = e->h_out[0];
    if (it_out) *it_out = it;
    if (lk_out) *lk_out = lk_last;
    if (reason_out) *reason_out = (int)e->h_out[6];
    if (history && it > 0) std::memcpy(history, e->h_hist, (size_t)it * sizeof(double));
    rc = done_with(VBNMF_OK);
    e->h_out[0] = lk_last;                                          // ml_likelihood() keeps answering for the pair held now
    return rc;
}

int vbnmf_engine_ml_get_state(vbnmf_engine *e, double *w, double *h)
{
    if (!e) return fail(VBNMF_ERR_BAD_ARG, "engine handle is NULL");
    if (!e->ml_ready) return fail(VBNMF_ERR_STATE, "ml_get_state before ml_set_state");
    if (int rc = use_device(e)) return rc;
    HIPCHECK(hipStreamSynchronize(e->stream));
    try {
        std::vector<double> tmp;
        if (w) {
            tmp.resize((size_t)e->n * e->R);
            HIPCHECK(hipMemcpy(tmp.data(), e->lw, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
            from_index_major(tmp, e->n, e->r, e->R, false, w);
        }
        if (h) {
            tmp.resize((size_t)e->m * e->R);
            HIPCHECK(hipMemcpy(tmp.data(), e->lh, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
            from_index_major(tmp, e->m, e->r, e->R, true, h, &e->cell_perm);
        }
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory staging the state");
    }
    return VBNMF_OK;
}

int vbnmf_engine_cluster_ids(vbnmf_engine *e, int32_t *ids)
{
    if (!e || !ids) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (!e->ml_ready && !e->has_state) return fail(VBNMF_ERR_STATE, "cluster_ids before a state was loaded");
    if (int rc = use_device(e)) return rc;
    const double *h = e->ml_ready ? e->lh : e->eh;   // ML: the coefficient matrix itself; VB: its posterior mean E[H]
    int32_t *d_ids = nullptr;
    if (int rc = dev_alloc(&d_ids, (size_t)e->m)) return rc;
    hipLaunchKernelGGL(k_argmax, dim3((unsigned)((e->m + 255) / 256)), dim3(256), 0, e->stream, h, e->m, e->r, e->R, d_ids);
    hipError_t he = hipGetLastError();
    std::vector<int32_t> stage;
    int32_t *host = ids;
    if (!e->cell_perm.empty()) { stage.resize((size_t)e->m); host = stage.data(); }      // labels arrive in the layout's cell order
    if (he == hipSuccess) he = hipMemcpyAsync(host, d_ids, (size_t)e->m * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    dev_free(d_ids);
    if (he != hipSuccess) return fail(VBNMF_ERR_HIP, "cluster_ids failed: %s", hipGetErrorString(he));
    if (!e->cell_perm.empty()) for (int64_t p = 0; p < e->m; p++) ids[e->cell_perm[p]] = stage[p];
    return VBNMF_OK;
}

// The connectivity stopping count of factorize() (reference R/factorize.R:198-208) on the device: the labels of the
// state held now against the labels of the previous call, as sum(cnn != cnn0) over all pairs of cells -- from the
// contingency table of the two labelings, never from the O(m^2) vectors.  changed = -1 on the first call after a
// state was loaded (the reference starts from nchange = npair, :200).  ids (optional): the new labels, m_local int32.
int vbnmf_engine_cluster_changes(vbnmf_engine *e, int64_t *changed, int32_t *ids)
{
    if (!e || !changed) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (!e->ml_ready && !e->has_state) return fail(VBNMF_ERR_STATE, "cluster_changes before a state was loaded");
    if (int rc = use_device(e)) return rc;
    for (int q = 0; q < 2; q++)
        if (!e->d_ids[q]) { if (int rc = dev_alloc(&e->d_ids[q], (size_t)e->m)) return rc; }
    const size_t tcount = (size_t)(e->r + 1) * (e->r + 1) + 1;
    if (!e->d_table) { if (int rc = dev_alloc(&e->d_table, tcount)) return rc; }
    const double *h = e->ml_ready ? e->lh : e->eh;
    const int cur = e->ids_cur ^ 1;
    hipLaunchKernelGGL(k_argmax, dim3((unsigned)((e->m + 255) / 256)), dim3(256), 0, e->stream, h, e->m, e->r, e->R, e->d_ids[cur]);
    HIPCHECK(hipGetLastError());
    unsigned long long n = 0;
    if (e->ids_valid) {
        HIPCHECK(hipMemsetAsync(e->d_table, 0, tcount * sizeof(unsigned long long), e->stream));
        hipLaunchKernelGGL(k_label_table, dim3((unsigned)((e->m + 255) / 256)), dim3(256), 0, e->stream, e->d_ids[cur ^ 1], e->d_ids[cur], e->m, e->r, e->d_table);
        HIPCHECK(hipGetLastError());
        hipLaunchKernelGGL(k_label_pairs, dim3(1), dim3(1), 0, e->stream, e->d_table, e->r, e->d_table + (tcount - 1));
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(&n, e->d_table + (tcount - 1), sizeof n, hipMemcpyDeviceToHost, e->stream));
    }
    std::vector<int32_t> stage;
    int32_t *host = ids;
    if (ids && !e->cell_perm.empty()) { stage.resize((size_t)e->m); host = stage.data(); }
    if (ids) HIPCHECK(hipMemcpyAsync(host, e->d_ids[cur], (size_t)e->m * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    if (ids && !e->cell_perm.empty()) for (int64_t p = 0; p < e->m; p++) ids[e->cell_perm[p]] = stage[p];
    *changed = e->ids_valid ? (int64_t)n : -1;
    e->ids_cur = cur;
    e->ids_valid = true;
    return VBNMF_OK;
}

}  // extern "C"

namespace {

// Sparse product with the resident X on operands that are ALREADY on the device: gene side C[n][R] = X B (B = the lh
// array, one row per cell), cell side C[m][R] = t(X) W (W = the lw array); the per-task partials are summed per
// major straight into `target`.
int spmm_device(vbnmf_engine *e, bool gene_side, double *target)
{
    SweepSide a = sweep_side_args(e, gene_side ? e->A : e->B, gene_side, gene_side ? e->epart : e->epart + e->n_wg);
    a.logterm = 0;
    a.stop = nullptr;
    int rc = VBNMF_ERR_BAD_ARG;
    switch (e->R) {
#define X(RR) case RR: rc = launch_spmm_r<RR>(e, a); break;
        VBNMF_FOR_EACH_R(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    if (rc) return rc;
    const DeviceSide &S = gene_side ? e->A : e->B;
    const int64_t cnt = (gene_side ? e->n : e->m) * e->R;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, S.part, S.inv_ptr, S.inv_task, gene_side ? e->n : e->m, e->R, target);
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// work space of the truncated SVD: [gram partials kGramBlocks * R*R | S (R*R) | S2 (R*R) | vals (R)]
struct SvdWs { double *gp, *S, *S2, *vals; };
SvdWs svd_ws(vbnmf_engine *e)
{
    const size_t RR = (size_t)e->R * e->R;
    SvdWs w;
    w.gp = e->svd_ws; w.S = w.gp + (size_t)kGramBlocks * RR; w.S2 = w.S + RR; w.vals = w.S2 + RR;
    return w;
}

int launch_gram(vbnmf_engine *e, const double *A, int64_t N)
{
    SvdWs w = svd_ws(e);
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_gram<RR>), dim3(kGramBlocks), dim3(1024), 0, e->stream, A, N, w.gp); break;
        VBNMF_FOR_EACH_R_UP_TO_64(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_small(vbnmf_engine *e, int mode, bool want_s2, double *vals_host, double seq)
{
    SvdWs w = svd_ws(e);
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_small<RR>), dim3(1), dim3(1024), 0, e->stream, w.gp, kGramBlocks, e->r, mode, w.S, want_s2 ? w.S2 : nullptr, w.vals, vals_host, seq, e->svd_status); break;
        VBNMF_FOR_EACH_R_UP_TO_64(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

int launch_apply(vbnmf_engine *e, const double *A, const double *S, int64_t N, double *B)
{
    switch (e->R) {
#define X(RR) case RR: hipLaunchKernelGGL((k_apply<RR>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, A, S, N, B); break;
        VBNMF_FOR_EACH_R_UP_TO_64(X)
#undef X
        default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
    }
    HIPCHECK(hipGetLastError());
    return VBNMF_OK;
}

// A <- orthonormal basis of range(A), A tall [N][R]: CholeskyQR twice (Gram matrix, its Cholesky factor's inverse,
// A times that) -- the second pass repairs the orthogonality the first loses to the condition number.  When `gram_ready`
// the block partials of t(A) A are already in the work space.
int orthonormalise(vbnmf_engine *e, double *A, int64_t N, bool gram_ready)
{
    SvdWs w = svd_ws(e);
    for (int pass = 0; pass < 2; pass++) {
        if (!(pass == 0 && gram_ready)) { if (int rc = launch_gram(e, A, N)) return rc; }
        if (int rc = launch_small(e, 0, false, nullptr, 0.0)) return rc;
        if (int rc = launch_apply(e, A, w.S, N, A)) return rc;
    }
    return VBNMF_OK;
}

}  // namespace

extern "C" {

// Truncated SVD of the resident X by block subspace iteration, ENTIRELY on the device (the
// irlba::irlba(mat, rank) of the svd2 initialiser, reference R/bayesian.R:150-159): the subspace has k = the engine's
// rank columns (rank_out of them are returned); per iteration two sparse products (k_spmm), two CholeskyQR2
// orthonormalisations (k_gram / k_small / k_apply) and the k x k eigen-problem that yields the current singular values
// (k_small, Jacobi) -- the host only reads those k values from pinned memory for the stopping rule
// max |s - s_old| <= tol * s[0] over the leading rank_out.  No host <-> device copy inside the iteration.
// Outputs: u (n x rank_out, column-major), d (rank_out), vt (rank_out x m, column-major), iterations done.
int vbnmf_engine_svd(vbnmf_engine *e, int32_t rank_out, double tol, int32_t maxit, uint64_t seed, double *u, double *d,
                     double *vt, int32_t *iterations)
{
    if (!e || !u || !d || !vt) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (e->partitioned) return fail(VBNMF_ERR_STATE, "the truncated SVD needs an unpartitioned engine");
    if (rank_out < 1 || rank_out > e->r) return fail(VBNMF_ERR_BAD_ARG, "rank_out must be in [1, engine rank]");
    if (e->R > VBNMF_MAX_SVD_COLUMNS)
        return fail(VBNMF_ERR_BAD_ARG, "the device-resident SVD holds at most %d subspace columns (engine rank %d); use the sparse products (vbnmf_engine_spmm) with a host QR", VBNMF_MAX_SVD_COLUMNS, e->r);
    if (maxit < 1) return fail(VBNMF_ERR_BAD_ARG, "maxit must be >= 1");
    if (int rc = use_device(e)) return rc;
    e->has_state = false; e->stats_ready = false; e->step_pending = false; e->prime_pending = false; e->ml_ready = false;
    const size_t RR = (size_t)e->R * e->R;
    if (!e->svd_ws) { if (int rc = dev_alloc(&e->svd_ws, (size_t)kGramBlocks * RR + 2 * RR + e->R)) return rc; }
    if (!e->svd_status) { if (int rc = dev_alloc(&e->svd_status, 1)) return rc; }
    HIPCHECK(hipMemsetAsync(e->svd_status, 0, sizeof(int32_t), e->stream));
    if (int rc = ensure_history(e, (size_t)e->R + 2)) return rc;             // pinned [R values | sequence number]
    SvdWs w = svd_ws(e);
    volatile double *hv = e->h_hist;
    hv[e->R] = 0.0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t nh = e->m * e->R;
    double *Q = e->lw, *Z = e->lh;                                           // n x k and m x k, index-major
    // range finder: Q = orth(X G), G standard normal m x k
    hipLaunchKernelGGL(k_normal_init, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, e->stream, Z, e->m, e->r, e->R, k0, k1);
    HIPCHECK(hipGetLastError());
    if (int rc = spmm_device(e, true, Q)) return rc;
    if (int rc = orthonormalise(e, Q, e->n, false)) return rc;
    std::vector<double> s_old;
    int it = 0;
    for (it = 1; it <= maxit; it++) {
        if (int rc = spmm_device(e, false, Z)) return rc;                    // Z = t(X) Q
        if (int rc = orthonormalise(e, Z, e->m, false)) return rc;
        if (int rc = spmm_device(e, true, Q)) return rc;                     // Y = X Z = Q Rm
        if (int rc = launch_gram(e, Q, e->n)) return rc;                     // t(Y) Y = t(Rm) Rm: eigenvalues = sigma^2
        if (int rc = launch_small(e, 1, false, e->h_hist_dev, (double)it)) return rc;
        if (int rc = orthonormalise(e, Q, e->n, true)) return rc;            // (the Gram partials are still in place)
        HIPCHECK(hipStreamSynchronize(e->stream));
        if (hv[e->R] != (double)it) return fail(VBNMF_ERR_HIP, "the singular values of iteration %d never arrived", it);
        std::vector<double> sv(rank_out);
        for (int q = 0; q < rank_out; q++) sv[q] = std::sqrt(std::max(0.0, (double)hv[q]));
        bool done = false;
        if (!s_old.empty()) {
            double dmax = 0.0;
            for (int q = 0; q < rank_out; q++) dmax = std::max(dmax, std::fabs(sv[q] - s_old[q]));
            done = dmax <= tol * sv[0];
        }
        s_old.swap(sv);
        if (done) break;
    }
    if (it > maxit) it = maxit;
    // B = t(Q) X = t(P), P = t(X) Q (m x k): B t(B) = t(P) P = Ub D^2 t(Ub); u = Q Ub, rows of vt = P Ub / D
    if (int rc = spmm_device(e, false, Z)) return rc;
    if (int rc = launch_gram(e, Z, e->m)) return rc;
    if (int rc = launch_small(e, 1, true, e->h_hist_dev, (double)(maxit + 1))) return rc;
    if (int rc = launch_apply(e, Q, w.S, e->n, e->ew)) return rc;
    if (int rc = launch_apply(e, Z, w.S2, e->m, e->eh)) return rc;
    HIPCHECK(hipStreamSynchronize(e->stream));
    int32_t status = 0;
    HIPCHECK(hipMemcpy(&status, e->svd_status, sizeof status, hipMemcpyDeviceToHost));
    if (status) return fail(VBNMF_ERR_STATE, "the subspace lost rank (a Gram matrix was not positive definite): fewer independent directions than the engine's rank");
    for (int q = 0; q < rank_out; q++) d[q] = std::sqrt(std::max(0.0, (double)hv[q]));
    try {
        std::vector<double> tmp((size_t)std::max(e->n, e->m) * e->R);
        HIPCHECK(hipMemcpy(tmp.data(), e->ew, (size_t)e->n * e->R * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < e->n; i++) for (int q = 0; q < rank_out; q++) u[i + (size_t)q * e->n] = tmp[(size_t)i * e->R + q];
        HIPCHECK(hipMemcpy(tmp.data(), e->eh, (size_t)e->m * e->R * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < e->m; j++) {
            const int64_t col = e->cell_perm.empty() ? j : e->cell_perm[j];
            for (int q = 0; q < rank_out; q++) vt[q + (size_t)col * rank_out] = tmp[(size_t)j * e->R + q];
        }
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory staging the singular vectors");
    }
    if (iterations) *iterations = it;
    return VBNMF_OK;
}

// ---------------------------------------------------------------- sparse products (truncated SVD of the svd2 initialiser)
int vbnmf_engine_spmm(vbnmf_engine *e, int32_t transpose, const double *B, double *C)
{
    if (!e || !B || !C) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (e->partitioned) return fail(VBNMF_ERR_STATE, "spmm needs an unpartitioned engine");
    if (int rc = use_device(e)) return rc;
    // the factor arrays serve as operand storage: whatever state the engine held is gone
    e->has_state = false; e->stats_ready = false; e->step_pending = false; e->prime_pending = false; e->ml_ready = false;
    const bool gene_side = transpose == 0;             // C = X t(B): lanes own genes and gather rows of B (one per cell)
    const int64_t n_in = gene_side ? e->m : e->n, n_out = gene_side ? e->n : e->m;
    double *operand = gene_side ? e->lh : e->lw;       // gathered through LDS
    double *dense = gene_side ? e->ew : e->eh;         // [n_out][R] result before the download
    try {
        std::vector<double> tmp;
        to_index_major(B, n_in, e->r, e->R, gene_side, tmp, gene_side ? &e->cell_perm : nullptr);     // B: one row per cell on the gene side
        HIPCHECK(hipMemcpyAsync(operand, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
        SweepSide a = sweep_side_args(e, gene_side ? e->A : e->B, gene_side, gene_side ? e->epart : e->epart + e->n_wg);
        a.logterm = 0;
        a.stop = nullptr;
        int rc = VBNMF_ERR_BAD_ARG;
        switch (e->R) {
#define X(RR) case RR: rc = launch_spmm_r<RR>(e, a); break;
            VBNMF_FOR_EACH_R(X)
#undef X
            default: return fail(VBNMF_ERR_BAD_ARG, "unsupported padded rank %d", e->R);
        }
        if (rc) return rc;
        const DeviceSide &S = gene_side ? e->A : e->B;
        const int64_t cnt = n_out * e->R;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, e->stream, S.part, S.inv_ptr, S.inv_task, n_out, e->R, dense);
        HIPCHECK(hipGetLastError());
        tmp.resize((size_t)cnt);
        HIPCHECK(hipMemcpyAsync(tmp.data(), dense, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
        from_index_major(tmp, n_out, e->r, e->R, !gene_side, C, gene_side ? nullptr : &e->cell_perm);
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory staging the operands");
    }
    return VBNMF_OK;
}

// ---------------------------------------------------------------- test hooks
int vbnmf_test_special_host(int32_t kind, int64_t n, const double *x, double *y)
{
    if (!x || !y || n < 0) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    for (int64_t i = 0; i < n; i++) {
        double psi, lg;
        switch (kind) {
            case 0: y[i] = dev_log(x[i]); break;
            case 1: dev_psi_lgamma(x[i], &psi, &lg); y[i] = psi; break;
            case 2: dev_psi_lgamma(x[i], &psi, &lg); y[i] = lg; break;
            case 6: { static LogTabEntry tab[kLogTabSize]; static bool init = false; if (!init) { fill_log_table(tab); init = true; } y[i] = dev_log_tab(x[i], tab); break; }
            default: y[i] = dev_div(1.0, x[i]); break;
        }
    }
    return VBNMF_OK;
}

int vbnmf_test_special_device(int32_t kind, int64_t n, const double *x, double *y)
{
    if (!x || !y || n < 0) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (int rc = check_device(0)) return rc;
    HIPCHECK(hipSetDevice(0));
    double *dx = nullptr, *dy = nullptr;
    LogTabEntry *dt = nullptr;
    std::vector<LogTabEntry> tab(kLogTabSize);
    fill_log_table(tab.data());
    if (int rc = dev_alloc(&dx, (size_t)n)) return rc;
    if (int rc = dev_alloc(&dy, (size_t)n)) { dev_free(dx); return rc; }
    if (int rc = dev_upload(&dt, tab)) { dev_free(dx); dev_free(dy); return rc; }
    int rc = VBNMF_OK;
    hipError_t he = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (he == hipSuccess && n > 0) {
        hipLaunchKernelGGL(k_test_special, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, kind, n, dx, dy, dt);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpy(y, dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (he != hipSuccess) rc = fail(VBNMF_ERR_HIP, "special-function test kernel failed: %s", hipGetErrorString(he));
    dev_free(dx); dev_free(dy); dev_free(dt);
    return rc;
}

// Test hook: a host function that sleeps `seconds` is enqueued on the engine's stream, so everything queued behind it
// waits that long without the GPU being busy or hung (tests/test_gpu_timeouts.py: the bounded waits).
static void sleep_host_fn(void *arg)
{
    const double s = *static_cast<double *>(arg);
    std::this_thread::sleep_for(std::chrono::duration<double>(s));
    delete static_cast<double *>(arg);
}

int vbnmf_test_stream_sleep(vbnmf_engine *e, double seconds)
{
    if (!e || !(seconds >= 0.0) || seconds > 60.0) return fail(VBNMF_ERR_BAD_ARG, "engine is NULL or seconds outside [0, 60]");
    if (int rc = use_device(e)) return rc;
    double *arg = new (std::nothrow) double(seconds);
    if (!arg) return fail(VBNMF_ERR_OOM, "out of host memory");
    hipError_t he = hipLaunchHostFunc(e->stream, sleep_host_fn, arg);
    if (he != hipSuccess) { delete arg; return fail(VBNMF_ERR_HIP, "hipLaunchHostFunc failed: %s", hipGetErrorString(he)); }
    return VBNMF_OK;
}

// ---------------------------------------------------------------- stateless forms
}  // extern "C"

namespace { uint64_t hash_bytes(const void *data, size_t bytes, uint64_t seed); }
extern "C" {
// Test hook (no device needed): the content hash of the stateless cache, for tests/test_cabi_symbols.py.
#ifdef VBNMF_ABL_STAMPS                                       /* instrumented builds only (profiles/ubench/r04/update_stamps.sh) */
int vbnmf_test_update_stamps(unsigned long long *out)
{
    if (hipDeviceSynchronize() != hipSuccess) return VBNMF_ERR_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_upd_stamps), sizeof(unsigned long long) * 2 * kUpdateBlocks * 12) == hipSuccess ? VBNMF_OK : VBNMF_ERR_HIP;
}
#endif
uint64_t vbnmf_test_hash_bytes(const void *data, int64_t bytes, uint64_t seed) { return (data && bytes >= 0) ? hash_bytes(data, (size_t)bytes, seed) : 0; }
}

namespace {

// The reference's caller hands the SAME X to every one of its thousands of calls (R/bayesian.R:339).  The stateless
// entries therefore keep the last ingested matrix and its engine alive, keyed by the CONTENT of X (dimensions + a
// 64-bit hash of every byte of the arrays handed in, formed by all host threads): a call with an X seen before costs
// the hash, the state upload, one step and the state download instead of ingestion + two layouts + an engine.
// VBNMF_STATELESS_CACHE=0 turns it off; vbnmf_stateless_cache_clear() releases what is held.
struct StatelessCache {
    std::mutex mu;                    // held for the whole call: stateless calls are serialised
    bool valid = false;
    int kind = 0;                     // 0 dense, 1 csc
    int64_t n = 0, m = 0, nnz = 0;
    uint64_t hash = 0, hash2 = 0;     // two independently seeded 64-bit hashes of the content: a 128-bit key
    std::vector<int32_t> colptr;      // csc: a copy of the pointer array, compared on a hit (not only hashed)
    std::vector<double> sample;       // dense: a strided sample of X (<= 4096 values), compared on a hit as well
    vbnmf_matrix *X = nullptr;
    vbnmf_engine *e = nullptr;
    int r = 0;
    void drop()
    {
        vbnmf_engine_destroy(e); e = nullptr;
        vbnmf_matrix_destroy(X); X = nullptr;
        valid = false;
        colptr.clear();
        sample.clear();
    }
};
StatelessCache &stateless_cache() { static StatelessCache c; return c; }

// Two independently seeded hashes of the same bytes in ONE pass over them (the stateless cache's 128-bit key).  Per 4 MB chunk
// FOUR multiply-xorshift chains per seed, fed the chunk's 8-byte words in turn (word q goes to chain q mod 4), so eight independent
// chains are in flight and a thread hashes at memory speed instead of at one multiply latency per word (a 3.7 MB dense matrix
// -- the reference's shipped data set -- used to take 0.7 ms of every literal drop-in call); chains and chunks are combined in
// order: the value does not depend on the thread count.  The seed starts EVERY chain of every chunk (not only the final combine),
// so two seeds give two INDEPENDENT 64-bit hashes: contents that collide in a chain under one seed do not under the other
// (tests/test_node_shared_cpu.py).  Small inputs are hashed by few threads (a thread is worth starting for a chunk or more).
void hash_bytes2(const void *data, size_t bytes, uint64_t seed_a, uint64_t seed_b, uint64_t &out_a, uint64_t &out_b)
{
    const size_t chunk = (size_t)1 << 22;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    std::vector<uint64_t> part_a(nchunks ? nchunks : 1, 0), part_b(nchunks ? nchunks : 1, 0);
    const unsigned char *p = static_cast<const unsigned char *>(data);
    constexpr uint64_t K = 0xFF51AFD7ED558CCDull, KL = 0xA24BAED4963EE407ull, KC = 0xC4CEB9FE1A85EC53ull;
    parallel_for((int64_t)nchunks, [&](int64_t b, int64_t e, int) {
        for (int64_t c = b; c < e; c++) {
            const size_t o = (size_t)c * chunk, len = std::min(chunk, bytes - o);
            const uint64_t salt = 0x9E3779B97F4A7C15ull ^ ((uint64_t)c * 0xD6E8FEB86659FD93ull);
            uint64_t ha[4], hb[4];
            for (int j = 0; j < 4; j++) { ha[j] = seed_a ^ salt ^ ((uint64_t)j * KL); hb[j] = seed_b ^ salt ^ ((uint64_t)j * KL); }
            size_t q = 0;
            for (; q + 32 <= len; q += 32) {
                uint64_t w[4];
                std::memcpy(w, p + o + q, 32);
                for (int j = 0; j < 4; j++) {
                    ha[j] = (ha[j] ^ w[j]) * K; hb[j] = (hb[j] ^ w[j]) * K;
                    ha[j] ^= ha[j] >> 32; hb[j] ^= hb[j] >> 32;
                }
            }
            for (int j = 0; q + 8 <= len; q += 8, j++) {
                uint64_t w;
                std::memcpy(&w, p + o + q, 8);
                ha[j] = (ha[j] ^ w) * K; hb[j] = (hb[j] ^ w) * K;
                ha[j] ^= ha[j] >> 32; hb[j] ^= hb[j] >> 32;
            }
            for (; q < len; q++) { ha[0] = (ha[0] ^ p[o + q]) * 0x100000001B3ull; hb[0] = (hb[0] ^ p[o + q]) * 0x100000001B3ull; }
            uint64_t da = ha[0], db = hb[0];
            for (int j = 1; j < 4; j++) { da = (da ^ ha[j]) * KC; da ^= da >> 29; db = (db ^ hb[j]) * KC; db ^= db >> 29; }
            part_a[c] = da; part_b[c] = db;
        }
    });
    uint64_t ha = seed_a, hb = seed_b;
    for (uint64_t v : part_a) { ha = (ha ^ v) * KC; ha ^= ha >> 29; }
    for (uint64_t v : part_b) { hb = (hb ^ v) * KC; hb ^= hb >> 29; }
    out_a = ha; out_b = hb;
}
uint64_t hash_bytes(const void *data, size_t bytes, uint64_t seed)
{
    uint64_t a, b;
    hash_bytes2(data, bytes, seed, ~seed, a, b);
    return a;
}

bool stateless_cache_enabled()
{
    const char *sv = getenv("VBNMF_STATELESS_CACHE");
    return !(sv && sv[0] == '0');
}

// The engine of rank r on the matrix with this key: from the cache, or ingested now through `ingest`.
int stateless_engine(int kind, int64_t n, int64_t m, int64_t nnz, uint64_t hash, uint64_t hash2, const int32_t *colptr, const double *dense, int32_t r,
                     const std::function<int(vbnmf_matrix **)> &ingest, vbnmf_engine **out)
{
    StatelessCache &C = stateless_cache();
    bool hit = C.valid && C.kind == kind && C.n == n && C.m == m && C.nnz == nnz && C.hash == hash && C.hash2 == hash2;
    if (hit && colptr) hit = C.colptr.size() == (size_t)m + 1 && std::memcmp(C.colptr.data(), colptr, ((size_t)m + 1) * sizeof(int32_t)) == 0;
    // dense: every 1/4096-th value (an odd stride, so the sample walks through all rows) beside the two hashes
    const size_t total = dense ? (size_t)n * (size_t)m : 0, stride = std::max<size_t>(1, total / 4096) | 1;
    if (hit && dense) {
        size_t q = 0;
        for (size_t o = 0; o < total && hit; o += stride, q++) hit = q < C.sample.size() && std::memcmp(&C.sample[q], dense + o, sizeof(double)) == 0;
        hit = hit && q == C.sample.size();
    }
    if (!hit) {
        C.drop();
        if (int rc = ingest(&C.X)) { C.X = nullptr; return rc; }
        C.kind = kind; C.n = n; C.m = m; C.nnz = nnz; C.hash = hash; C.hash2 = hash2; C.valid = true;
        if (colptr) C.colptr.assign(colptr, colptr + m + 1);
        if (dense) for (size_t o = 0; o < total; o += stride) C.sample.push_back(dense[o]);
    }
    // The stateless entries take no device argument (the reference's signature has none): VBNMF_DEVICE names it, read at
    // every call, so the slaves of Rmpi::mpi.applyLB (reference R/bayesian.R:262-263) that call the plain shim can each be
    // given their own GPU; default device 0.
    int device = 0;
    if (const char *dv = getenv("VBNMF_DEVICE")) device = atoi(dv);
    if (!C.e || C.r != r || C.e->device != device) {
        vbnmf_engine_destroy(C.e); C.e = nullptr;
        if (int rc = vbnmf_engine_create(C.X, r, device, &C.e)) { C.e = nullptr; return rc; }
        C.r = r;
    }
    *out = C.e;
    return VBNMF_OK;
}

int update_once(vbnmf_engine *e, const double *lw_in, const double *lh_in, const double *eh_in,
                double aw, double bw, double ah, double bh, double fudge,
                double *lw, double *lh, double *ew, double *eh, double *dw, double *dh, double *lkh)
{
    int rc = vbnmf_engine_set_state(e, lw_in, lh_in, eh_in);
    if (!rc) rc = vbnmf_engine_step(e, aw, bw, ah, bh, fudge, lkh, nullptr);
    if (!rc) rc = vbnmf_engine_get_state(e, lw, lh, ew, eh, dw, dh);
    return rc;
}

int ml_update_once(vbnmf_engine *e, const double *w_in, const double *h_in, int32_t prior, double gamma_a, double gamma_b,
                   double *w, double *h, double *lk)
{
    int rc = vbnmf_engine_ml_set_state(e, w_in, h_in);
    if (!rc) rc = vbnmf_engine_ml_step(e, prior, gamma_a, gamma_b, lk);
    if (!rc) rc = vbnmf_engine_ml_get_state(e, w, h);
    return rc;
}

// dense / CSC front ends shared by the VB and ML stateless calls: `use(engine)` runs under the cache's lock
int with_dense(int64_t n, int64_t m, int32_t r, const double *X, const std::function<int(vbnmf_engine *)> &use)
{
    if (n <= 0 || m <= 0 || !X) return fail(VBNMF_ERR_BAD_ARG, "X is NULL or has a non-positive dimension");
    StatelessCache &C = stateless_cache();
    std::lock_guard<std::mutex> g(C.mu);
    const bool cache = stateless_cache_enabled();
    uint64_t h = 0, h2 = 0;
    if (cache) hash_bytes2(X, (size_t)n * (size_t)m * sizeof(double), 0x64656E7365ull, 0x3243F6A8885A308Dull, h, h2);
    if (!cache) C.drop();
    vbnmf_engine *e = nullptr;
    int rc = stateless_engine(0, n, m, 0, h, h2, nullptr, X, r, [&](vbnmf_matrix **M) { return vbnmf_matrix_from_dense(n, m, X, M); }, &e);
    if (!rc) rc = use(e);
    if (!cache || rc) C.drop();
    return rc;
}

int with_csc(int64_t n, int64_t m, int32_t r, const int32_t *p, const int32_t *i, const double *x,
             const std::function<int(vbnmf_engine *)> &use)
{
    if (n <= 0 || m <= 0 || !p) return fail(VBNMF_ERR_BAD_ARG, "a CSC slot pointer is NULL or a dimension is non-positive");
    StatelessCache &C = stateless_cache();
    std::lock_guard<std::mutex> g(C.mu);
    const bool cache = stateless_cache_enabled();
    // the pointer array is checked BEFORE anything is read through it (the hash below walks nnz elements of i and x)
    if (p[0] != 0) return fail(VBNMF_ERR_BAD_ARG, "pointer array must start at 0");
    for (int64_t j = 0; j < m; j++)
        if (p[j + 1] < p[j]) return fail(VBNMF_ERR_BAD_ARG, "pointer array is not non-decreasing at %lld", (long long)j);
    const int64_t nnz = p[m];
    if (nnz > 0 && (!i || !x)) return fail(VBNMF_ERR_BAD_ARG, "a CSC slot pointer is NULL");
    uint64_t h = 0, h2 = 0;
    if (cache) {
        hash_bytes2(i, (size_t)nnz * sizeof(int32_t), 0x637363ull, 0x3243F6A8885A308Dull, h, h2);
        hash_bytes2(x, (size_t)nnz * sizeof(double), h, h2, h, h2);
    }
    if (!cache) C.drop();
    vbnmf_engine *e = nullptr;
    int rc = stateless_engine(1, n, m, nnz, h, h2, p, nullptr, r, [&](vbnmf_matrix **M) { return vbnmf_matrix_from_csc(n, m, p, i, x, M); }, &e);
    if (!rc) rc = use(e);
    if (!cache || rc) C.drop();
    return rc;
}

}  // namespace

extern "C" {

// Gives every device buffer the pool holds back to the driver (buffers in use by live engines are not touched).
void vbnmf_pool_trim(void)
{
    DevicePool &P = device_pool();
    std::vector<PoolBuf> drop;
    { std::lock_guard<std::mutex> g(P.mu); drop.swap(P.free_list); P.held = 0; }
    for (const PoolBuf &b : drop) (void)hipFree(b.p);
}

void vbnmf_stateless_cache_clear(void)
{
    StatelessCache &C = stateless_cache();
    std::lock_guard<std::mutex> g(C.mu);
    C.drop();
}

int vbnmf_update_dense(int64_t n, int64_t m, int32_t r, const double *X, const double *lw_in, const double *lh_in,
                       const double *eh_in, double aw, double bw, double ah, double bh, double fudge,
                       double *lw, double *lh, double *ew, double *eh, double *dw, double *dh, double *lkh)
{
    if (!lw_in || !lh_in || !eh_in) return fail(VBNMF_ERR_BAD_ARG, "a wh member is NULL");
    return with_dense(n, m, r, X, [&](vbnmf_engine *e) {
        return update_once(e, lw_in, lh_in, eh_in, aw, bw, ah, bh, fudge, lw, lh, ew, eh, dw, dh, lkh); });
}

int vbnmf_update_csc(int64_t n, int64_t m, int32_t r, const int32_t *p, const int32_t *i, const double *x,
                     const double *lw_in, const double *lh_in, const double *eh_in,
                     double aw, double bw, double ah, double bh, double fudge,
                     double *lw, double *lh, double *ew, double *eh, double *dw, double *dh, double *lkh)
{
    if (!lw_in || !lh_in || !eh_in) return fail(VBNMF_ERR_BAD_ARG, "a wh member is NULL");
    return with_csc(n, m, r, p, i, x, [&](vbnmf_engine *e) {
        return update_once(e, lw_in, lh_in, eh_in, aw, bw, ah, bh, fudge, lw, lh, ew, eh, dw, dh, lkh); });
}

int vbnmf_ml_update_dense(int64_t n, int64_t m, int32_t r, const double *X, const double *w_in, const double *h_in,
                          int32_t prior, double gamma_a, double gamma_b, double *w, double *h, double *lk)
{
    if (!w_in || !h_in) return fail(VBNMF_ERR_BAD_ARG, "w or h is NULL");
    return with_dense(n, m, r, X, [&](vbnmf_engine *e) { return ml_update_once(e, w_in, h_in, prior, gamma_a, gamma_b, w, h, lk); });
}

int vbnmf_ml_update_csc(int64_t n, int64_t m, int32_t r, const int32_t *p, const int32_t *i, const double *x,
                        const double *w_in, const double *h_in, int32_t prior, double gamma_a, double gamma_b,
                        double *w, double *h, double *lk)
{
    if (!w_in || !h_in) return fail(VBNMF_ERR_BAD_ARG, "w or h is NULL");
    return with_csc(n, m, r, p, i, x, [&](vbnmf_engine *e) { return ml_update_once(e, w_in, h_in, prior, gamma_a, gamma_b, w, h, lk); });
}

}  // extern "C"
