// common.h -- shared host-side declarations of libvbnmf_hip.so (not part of the C ABI).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vbnmf.h"

namespace vbnmf {

// ---- error plumbing: one message per host thread, surfaced by vbnmf_last_error() ----
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

// ---- simple fork-join over [0, count) with std::thread (no OpenMP runtime needed) ----
void parallel_for(int64_t count, const std::function<void(int64_t begin, int64_t end, int tid)> &fn,
                  int max_threads = 0);
int host_threads();
void set_host_threads_override(int n);      // 0: back to the default rule (cores of the affinity mask, at most 32)
void set_thread_share(int share);           // the CALLING thread's parallel_for calls default to host_threads() / share threads (1: all)

// The same matrix by rows (genes), columns ascending in each row: what the gene side of the layout is cut from.
// std::allocator whose resize() leaves trivially constructible elements uninitialised (no zero-fill pass over the
// 200 MB entry arrays: the builder writes every slot itself, and the pages are first touched by the threads that fill them).
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using BigVec = std::vector<T, NoInitAlloc<T>>;

struct RowMajor {
    std::vector<int64_t> ptr;      // n+1
    BigVec<int32_t> idx;           // nnz
    BigVec<double> val;            // nnz
};
struct RowMajorCache {
    std::once_flag once;
    RowMajor rm;
};
// The layout-internal renumbering of the cells of the WHOLE matrix (order.cpp): perm[p] = the original column stored
// at position p; empty = identity.  Computed once per matrix (a shell takes it from the first imported layout).
struct CellOrderCache {
    std::once_flag once;
    std::vector<int32_t> perm;
};

// ---- canonical host copy of X: CSC, rows ascending in each column, no zeros, no dups ----
struct Matrix {
    int64_t n = 0, m = 0, nnz = 0;
    std::vector<int64_t> colptr;   // m+1
    std::vector<int32_t> row;      // nnz
    std::vector<double> val;       // nnz
    std::shared_ptr<RowMajorCache> rm_cache = std::make_shared<RowMajorCache>();
    const RowMajor &row_major() const;   // lazily built, thread-safe; engines of every rank on this matrix share it.
                                         // Column ids are POSITIONS in cell_order() (the gene side's minors).
    std::shared_ptr<CellOrderCache> order_cache = std::make_shared<CellOrderCache>();
    const std::vector<int32_t> &cell_order() const;   // whole matrix; lazily computed, thread-safe
    bool shell = false;            // metadata only (vbnmf_matrix_shell): no entries; layouts come from vbnmf_matrix_import_layout
    bool counts_u16 = false;       // every stored value is an integer in [1, kPackedCountMax]
    bool counts_int = false;       // every stored value is an integer in [1, 2^31): the 4-byte entry format applies
                                   // (a count above kPackedCountMax is stored as several entries of the same minor --
                                   // every use of an entry in the sweeps is linear in its value)
    double max_val = 0.0;
};

// sum over stored entries of lgamma(x+1) for columns [cb, ce), fixed summation order.
double sum_lgamma_x1(const Matrix &X, int64_t cb, int64_t ce);
// sum over stored entries of -x log x + x (the constant of the ML-NMF likelihood, reference R/factorize.R:46-47).
double sum_xlogx(const Matrix &X, int64_t cb, int64_t ce);

// ---- tiled device layout of one side (DESIGN.md "Data layout in HBM") ----
//
// A *task* is a run of at most `max_len` stored entries of one major (gene on side 0, cell
// on side 1) whose minors fall in one minor block; one lane of the sweep kernel owns it.
// Tasks of a block are sorted by length (descending) and cut into *slices* of 64 (one
// wavefront).  Each block's slices are dealt out in strides so that any contiguous stretch of
// the resulting work list holds a mix of long and short slices; the list is cut into `n_wg`
// stretches of equal cost, one per persistent workgroup; a *segment* is the part of a
// stretch that lies in one block (the workgroup stages that block of the gathered factor in
// LDS once per segment).  Slices are numbered in processing order -- workgroup by workgroup,
// segment by segment, longest first inside a segment -- and the waves of the workgroup pull
// them through a ticket counter, so a segment is just the id range [seg_ptr[g], seg_ptr[g+1]).
constexpr int kLanes = 64;          // one slice = one wavefront
constexpr int kUnroll = 4;          // entries per lane per 16-byte load
constexpr int kWidthQuantum = 4;    // slice widths are multiples of this (one 4-entry group of the packed stream)
constexpr uint32_t kIdleLane = 0xFFFFFFFFu;
// Packed entry word: bits 4..17 = the LDS slot (16-byte unit) of the minor's factor row, already shifted into a
// byte offset (word & kPackedOffsetMask); bits 18..31 = the count.  Padding slots are 0 (row 0, count 0).
constexpr int kPackedCountShift = 18;
constexpr uint32_t kPackedOffsetMask = 0x3FFF0u;
constexpr double kPackedCountMax = 16383.0;

// The big arrays of a layout (the packed entry stream: 200 MB a side at the headline size) live either in the library's
// own memory or INSIDE a shared-memory segment mapped by every process of the node (vbnmf_matrix_share_layout /
// _attach_layout): the builder writes them there as it fills them, the peers map them -- one copy of the layout in host
// memory per node, no export / import pass.  `keep` holds the mapping for as long as the layout lives.
template <class T>
struct ExtVec {
    using value_type = T;
    T *p = nullptr;
    size_t n = 0;
    BigVec<T> own;
    std::shared_ptr<void> keep;
    ExtVec() = default;
    ExtVec(const ExtVec &) = delete;
    ExtVec &operator=(const ExtVec &) = delete;
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void resize(size_t k) { keep.reset(); own.resize(k); p = own.data(); n = k; }
    void adopt(T *q, size_t k, std::shared_ptr<void> holder) { BigVec<T>().swap(own); p = q; n = k; keep = std::move(holder); }
};

struct Layout {
    int side = 0;                   // 0: lanes own genes, minors = cells; 1: lanes own cells, minors = genes
    bool wide = false;
    int64_t n_major = 0, n_minor = 0;
    int32_t block_width = 0, n_blocks = 0, max_len = 0, n_wg = 0, row_slots = 0;   // block_width: the WIDEST block
    int64_t n_tasks = 0, n_slices = 0, n_slots = 0, n_segs = 0, nnz = 0;
    std::vector<uint32_t> task_major;    // n_slices * 64 ; kIdleLane pads a block's last slice
    std::vector<int32_t> slice_width;    // n_slices ; entries per lane, multiple of 4
    std::vector<int64_t> slice_off;      // n_slices ; first slot of the slice
    std::vector<int32_t> slice_block;    // n_slices ; minor block (host-side bookkeeping / tests)
    std::vector<int32_t> slice_fast;     // n_slices ; leading entries per lane that are stored ones in EVERY lane (multiple of 8)
    std::vector<int64_t> block_start;    // n_blocks + 1 ; first minor of each block (blocks differ in width, see build_layout)
    std::vector<int32_t> seg_block;      // n_segs
    std::vector<int32_t> wg_seg0;        // n_wg + 1
    std::vector<int32_t> seg_ptr;        // n_segs + 1 : first slice of each segment
    std::vector<int32_t> inv_ptr;        // n_major + 1 : tasks of each major ...
    std::vector<uint32_t> inv_task;      // n_tasks     : ... as slice*64+lane ids, in (block, position) order
    std::vector<int32_t> cell_perm;      // cells of the column range in layout order: position -> original local column
                                         // (minors on side 0, majors on side 1; empty = identity).  order.cpp
    ExtVec<uint32_t> packed;             // n_slots (wide == false)
    ExtVec<uint32_t> wide_idx;           // n_slots (wide == true)
    ExtVec<double> wide_val;             // n_slots (wide == true)
};

// Where build_layout puts the big arrays: called once the small arrays are final and n_slots is known (null: the
// library's own memory).  place() must make L.packed (or L.wide_idx / L.wide_val) n_slots long.
struct LayoutSink {
    virtual ~LayoutSink() = default;
    virtual int place(Layout &L) = 0;
};

struct LayoutParams {
    int32_t block_width;   // minors per LDS block when the blocks are cut equal
    int32_t block_cap;     // most minors a block may hold (LDS capacity at this rank); 0 = block_width
    int32_t max_len;       // longest task (entries), multiple of 4
    int32_t n_wg;          // persistent workgroups of the sweep kernel
    int32_t row_slots;     // 16-byte LDS slots per staged factor row at this rank (lds_row_bytes / 16)
};

// Padded rank used on the device (even, so a factor row is a whole number of 16-byte LDS reads).
// Ranks up to 32 are padded to even; above, SP = 2 (to 64) or 4 (to 128) lanes of the sweep share a task's R * SP
// columns (kernels.h: sweep_side), R a multiple of 4 between 20 and 32: multiples of 8, then of 16.
inline int padded_rank(int r) { return r <= 32 ? (r + 1) & ~1 : (r <= 64 ? (r + 7) & ~7 : (r + 15) & ~15); }
constexpr int rank_shares(int RT) { return RT <= 32 ? 1 : (RT <= 64 ? 2 : 4); }
// Bytes of one factor row in the sweep's LDS image: R doubles, padded to an odd number of 16-byte
// bank slots so that rows congruent mod 16 (and only those) start in the same slot.
constexpr int lds_row_bytes(int R) { return ((R / 2) | 1) * 16; }
// LDS the sweep keeps for itself in front of the factor block: the 128-entry ln table (2048 B),
// 128 per-slice evidence partials (1024 B; 256 until round 5: a segment holds 24-35 slices at the headline size, and 1 KB more of
// block is twelve more factor rows -- 20 000 genes now fall in 10 blocks instead of 11, 50 000 cells in 25 instead of 26) and the slice
// ticket counter (16 B).
constexpr int kLdsReserveBytes = 3088;
// Threads per workgroup of the sweep kernel at padded rank R: as many waves per SIMD as the
// kernel's register need (factor row + accumulators + two gathered rows, ~14 R + 20 VGPRs) allows
// without spilling: 4 waves/SIMD up to R = 4, 3 up to 14 (measured on the C3 matrix, round 2: 768 against 512 threads
// 0.220 / 0.225 ms at rank 12, 0.250 / 0.253 at 14, 0.309 / 0.303 at 16), 2 beyond.  From R = 28 on the loop keeps ONE
// gathered-row buffer (factor row + accumulators + one row = 6 R VGPRs; two rows spill at 2 waves), which also fits 2
// waves per SIMD: sweep 0.762 -> 0.695 ms at rank 28 and 1.132 -> 0.873 ms at rank 32 against 1 wave (C3 matrix).
#ifndef VBNMF_T768_UPTO
#define VBNMF_T768_UPTO 14          // largest padded rank run with 768 threads (3 waves / SIMD); experiments override it
#endif
#ifndef VBNMF_ONEBUF_FROM
#define VBNMF_ONEBUF_FROM 28        // smallest padded rank whose sweep keeps ONE gathered-row buffer (no LDS look-ahead)
#endif
#ifndef VBNMF_T512_UPTO
#define VBNMF_T512_UPTO 32          // largest padded rank run with 512 threads (2 waves / SIMD)
#endif
#ifndef VBNMF_T1024_UPTO
#define VBNMF_T1024_UPTO 4          // largest padded rank run with 1024 threads (4 waves / SIMD)
#endif
#ifndef VBNMF_ONEBUF_UPTO
#define VBNMF_ONEBUF_UPTO 0         // experiments: padded ranks <= this ALSO use the one-buffer loop
#endif
constexpr int sweep_threads_lane(int R) { return R <= VBNMF_T1024_UPTO ? 1024 : (R <= VBNMF_T768_UPTO ? 768 : (R <= VBNMF_T512_UPTO ? 512 : 256)); }
// by the padded rank RT: the geometry of the per-lane rank RT / shares
constexpr int sweep_threads(int RT) { return sweep_threads_lane(RT / rank_shares(RT)); }
// Default block width / task length for a side at padded rank R with `nnz` stored entries (0: unknown, longest
// tasks); n_wg <= 0 picks the default (256).
LayoutParams default_layout_params(int64_t n_major, int64_t n_minor, int R, int n_wg = 0, int64_t nnz = 0);

// Build the layout of `side` for columns [cb, ce) of X, the cells renumbered by `perm` (position -> local column; null or
// empty: as stored).  The whole matrix (cb = 0, ce = m) must be given X.cell_order(): the cached row-major copy is in it.
int build_layout(const Matrix &X, int64_t cb, int64_t ce, int side, const LayoutParams &lp, const std::vector<int32_t> *perm, Layout &out,
                 LayoutSink *sink = nullptr);
// order.cpp: the renumbering for columns [cb, ce) (empty: identity / switched off: VBNMF_CELL_ORDER=0; =1 forces it on
// for every size; default: sparse matrices with at least 8192 cells in the range).
std::vector<int32_t> compute_cell_order(const Matrix &X, int64_t cb, int64_t ce);

const char *last_error_cstr();

// Canonical matrix from compressed columns in any order within a column (duplicates summed, zeros dropped).
int matrix_from_csc(int64_t n, int64_t m, const int32_t *p, const int32_t *i, const double *x, Matrix &X);

}  // namespace vbnmf

// opaque handles of the C ABI
// Whole-matrix layouts already cut for this matrix, by geometry.  The tiled layout depends on the rank only through
// the LDS row size (the same for padded ranks 8 and 10, 12 and 14, ...), so the engines of a rank sweep and the
// restarts of a rank share a few of them instead of rebuilding one each (seconds of host time at C3).
struct LayoutCache {
    struct Entry {
        int side;
        vbnmf::LayoutParams lp;
        std::shared_ptr<const vbnmf::Layout> layout;
    };
    // The device-resident copy of a cached layout's arrays on one device, shared by every engine created from it
    // there (engine.hip owns the type behind `arrays`; it is freed when the last engine and the cache let go).
    struct DeviceCopy {
        const vbnmf::Layout *key;
        int device;
        std::shared_ptr<void> arrays;
    };
    std::mutex mu;
    std::vector<Entry> entries;    // most recent last; capped (VBNMF_LAYOUT_CACHE pairs, default 3, 0 = off)
    std::vector<DeviceCopy> copies;  // only of layouts still in `entries`
};
struct vbnmf_matrix {
    vbnmf::Matrix M;
    double lgx = 0.0;      // sum over stored entries of lgamma(x+1)
    mutable LayoutCache layouts;
    // Rank classes (vbnmf_matrix_plan_ranks): padded ranks, ascending.  An engine of padded rank R on this matrix takes the
    // geometry (LDS block width, row stride) of the smallest class >= R -- its own when there is none.
    mutable std::mutex plan_mu;
    mutable std::vector<int32_t> plan;
    // sum over stored entries of -x log x + x, formed once (whole matrix; ML-NMF likelihood constant)
    mutable std::once_flag xlx_once;
    mutable double xlx = 0.0;
    // vbnmf_matrix_prepare_async: the cell order and the row-major copy being formed on a background host thread
    mutable std::mutex prep_mu;
    mutable std::thread prep;
    ~vbnmf_matrix() { if (prep.joinable()) prep.join(); }
};
namespace vbnmf {
// The layout of `side` for the whole matrix at the default geometry of padded rank R: from the matrix's cache, or
// built now (and cached).  rc != 0 and a null pointer on failure.
std::shared_ptr<const Layout> shared_layout(const vbnmf_matrix *X, int side, const LayoutParams &lp, int &rc, LayoutSink *sink = nullptr,
                                            bool *built = nullptr);
// Device copies of cached layouts: look one up (null if absent) / remember one (ignored if the layout is not cached).
std::shared_ptr<void> cached_device_copy(const vbnmf_matrix *X, const Layout *L, int device);
void store_device_copy(const vbnmf_matrix *X, const Layout *L, int device, std::shared_ptr<void> arrays);
// The padded rank whose geometry an engine of padded rank R uses on this matrix (R itself without a plan).
int plan_class(const vbnmf_matrix *X, int R);
// Rank classes of a sweep (padded ranks, ascending): the largest planned rank, then up to max_classes - 1 further ones,
// each the largest planned rank whose LDS rows are at most half as wide as the previous class's.
std::vector<int32_t> rank_classes(const int32_t *ranks, int32_t count, int32_t max_classes);
// Persistent workgroups of the sweep on `device` (one per CU; VBNMF_NWG overrides; a partition's sweep leaves
// VBNMF_COMM_CUS free for the all-reduce kernels).  rc != 0 on a HIP error.  Defined in engine.hip.
int sweep_workgroups(int device, bool partitioned, int &n_wg);
// Adds an already built layout to the matrix's cache (imported from another process of the node).
void cache_layout(const vbnmf_matrix *X, int side, const LayoutParams &lp, std::shared_ptr<const Layout> L);
}
struct vbnmf_layout {
    vbnmf::Layout L;
};

namespace vbnmf {
// Allocates the handle, runs `fill` on its matrix and finishes it (sum lgamma(x+1)); maps exceptions to codes.
int new_matrix(vbnmf_matrix **out, const std::function<int(Matrix &)> &fill);
}
