// common.h -- shared host-side declarations of libvbnmf_hip.so (not part of the C ABI).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
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

// ---- canonical host copy of X: CSC, rows ascending in each column, no zeros, no dups ----
struct Matrix {
    int64_t n = 0, m = 0, nnz = 0;
    std::vector<int64_t> colptr;   // m+1
    std::vector<int32_t> row;      // nnz
    std::vector<double> val;       // nnz
    bool counts_u16 = false;       // every stored value is an integer in [1, 65535]
};

// sum over stored entries of lgamma(x+1) for columns [cb, ce), fixed summation order.
double sum_lgamma_x1(const Matrix &X, int64_t cb, int64_t ce);

// ---- tiled device layout of one side (DESIGN.md "Data layout in HBM") ----
constexpr int kLanes = 64;          // one slice = one wavefront
constexpr int kUnroll = 4;          // entries per lane per 16-byte load
constexpr uint32_t kIdleLane = 0xFFFFFFFFu;

struct Layout {
    int side = 0;                   // 0: lanes own genes, minors = cells; 1: lanes own cells, minors = genes
    bool wide = false;
    int64_t n_major = 0, n_minor = 0;
    int32_t block_width = 0, n_blocks = 0, chunk = 0;
    int64_t n_tiles = 0, n_slices = 0, n_slots = 0, nnz = 0;
    std::vector<int32_t> tile_block;
    std::vector<int64_t> tile_slice0;    // n_tiles + 1
    std::vector<uint32_t> slice_major;   // n_slices * 64
    std::vector<int32_t> slice_width;    // n_slices
    std::vector<int64_t> slice_off;      // n_slices
    std::vector<uint32_t> packed;        // n_slots (wide == false)
    std::vector<uint32_t> wide_idx;      // n_slots (wide == true)
    std::vector<double> wide_val;        // n_slots (wide == true)
};

struct LayoutParams {
    int32_t block_width;   // minors per LDS block
    int32_t chunk;         // majors per tile, multiple of 64
};

// Padded rank used on the device (even, so a factor row is a whole number of 16-byte LDS reads).
inline int padded_rank(int r) { return (r + 1) & ~1; }
// Threads per workgroup of the sweep kernel at padded rank R (register budget: 128 VGPRs at 1024, 256 at 512).
constexpr int sweep_threads(int R) { return R <= 4 ? 1024 : 512; }
// Default block width / chunk for a side at padded rank R (LDS budget, tile count).
LayoutParams default_layout_params(int64_t n_major, int64_t n_minor, int R);

// Build the layout of `side` for columns [cb, ce) of X.
int build_layout(const Matrix &X, int64_t cb, int64_t ce, int side, const LayoutParams &lp, Layout &out);

const char *last_error_cstr();

}  // namespace vbnmf

// opaque handles of the C ABI
struct vbnmf_matrix {
    vbnmf::Matrix M;
    double lgx = 0.0;      // sum over stored entries of lgamma(x+1)
};
struct vbnmf_layout {
    vbnmf::Layout L;
};
