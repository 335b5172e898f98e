// kernels.h -- hand-written gfx950 kernels of the VB-NMF step (included once, by engine.hip).
//
// Device state layout (all fp64, "index-major": one factor row = R contiguous doubles,
// R = rank padded to even, pad columns hold 0):
//   lw, llw, ew, dw : [n][R]       llw = lw * log(lw)
//   lh, llh, eh, dh : [m][R]       llh = lh * log(lh)
//   part{A,B}       : [n_slices*64][R]  per-task partial statistics (task = one major's entries in one minor block),
//                     summed per major in the fixed order of the layout's inverse index
//
// Mathematics (reference src/vbnmf_update.cpp, all citations to that file):
//   sweep     :33-36  wth_ij = sum_k lw_ik lh_kj ; q_ij = X_ij / wth_ij ;
//                     accA_ik = sum_j q_ij lh_kj  (sw = lw .* accA) ; accB_kj = sum_i lw_ik q_ij
//             :67-77  the data part of the evidence, using on the SAME (new) lw, lh the identity
//                     sum_ij X_ij (A+B)_ij / wth_ij = sum_ik sw_ik log lw_ik + sum_kj sh_kj log lh_kj
//                     (A = (lw.*log lw) lh, B = lw (lh.*log lh)), so one pass over X yields the
//                     statistics of step t+1 and the evidence of step t.
//   update    :38-65  Gamma posterior shape/rate, means, variances, geometric means;
//             :82-89  the prior/entropy terms of the evidence.
//   final     :78,90  -sum(ew*eh) = -sum_k colSum(ew)_k rowSum(eh)_k ; U / (n*m).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "special.h"

namespace vbnmf {

constexpr uint32_t kIdle = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------
// Sweep.  One persistent workgroup per CU; each walks its cost-balanced share of the slices, first of
// the gene side then of the cell side.  Within a segment (its slices of one minor block) that
// block of the gathered factor G sits in LDS and the waves pull slices, longest first, through an
// LDS ticket counter; each lane owns one task (a run of one major's entries): the major's factor
// row F and R accumulators live in VGPRs, the entries are stored lane-interleaved so a wave reads
// 1 KiB per load instruction, and the partial statistics are written task-major (a wave writes one
// contiguous 64*R*8-byte piece).  No floating-point atomics anywhere: results are bit-reproducible.
// ------------------------------------------------------------------------------------
struct SweepSide {
    const uint32_t *packed;        // (count << 18) | (LDS slot of the minor's row << 4)   (packed layout, common.h)
    const uint32_t *widx;          // local minor                           (wide layout)
    const double *wval;            // value                                 (wide layout)
    const uint32_t *task_major;    // [n_slices][64]
    const int32_t *slice_width;    // [n_slices]
    const int32_t *slice_fast;     // [n_slices] low 16 bits: leading entries per lane that are stored ones in every lane
                                   // (multiple of 8); high 16 bits: ... that are ones or twos in every lane (>= the low half)
    const int64_t *slice_off;      // [n_slices]
    const int32_t *seg_block;      // [n_segs]
    const int32_t *block_start;    // [n_blocks + 1] first minor of each block
    const int32_t *wg_seg0;        // [n_wg + 1]
    const int32_t *seg_ptr;        // [n_segs + 1] first slice of each segment (slices are numbered in processing order)
    const double *F;               // [n_major][R]  factor owned by the lanes
    const double *llF;             // [n_major][R]  F * log F
    const double *G;               // [n_minor][R]  factor gathered through LDS
    double *part;                  // [n_slices*64][R] partial statistics per task
    double *epart;                 // [n_wg] evidence partials, one per workgroup
    double *csl;                   // [n_slices][RT] per-slice column sums of F .* acc (gene side of the VB sweep; null: not formed) ...
    double *csum;                  // ... and [n_wg][RT] their per-workgroup sums, in list order: sum_i sw_ik without the W update
                                   // having run (k_update2: both posterior updates in one launch)
    int32_t n_minor;
    int32_t row_slots;             // LDS row stride of the staged factor block in 16-byte slots (odd; >= R / 2: the stride of the
                                   // layout's rank class, which may be wider than this rank's own rows -- common.h, rank classes)
    int32_t logterm;               // this side also accumulates sum x*log(wth)
    int32_t n_wg;
    const LogTabEntry *logtab;     // [128] ln table (staged at the front of LDS)
    int32_t pull_ends;             // the youngest waves pull slices from the short end of a segment's list (take_ticket_ends)
    int32_t stream_nt;             // the entry stream is read with the non-temporal policy (it does not fit the Infinity Cache)
    const int32_t *stop;           // device-driven loop: the sweep returns at once when *stop != 0 (or null)
    unsigned long long *dbg;       // diagnostic: [n_wg][2 + 2*waves] 100 MHz timestamps, or null
};

template <int R>
struct SweepRegs {
    double F[R];
    double acc[R];
    double lsum;
    double prod;                   // running product of wth over the slice's leading ones, kept in [0.25, 1) ...
    int pexp;                      // ... with its binary exponent here: sum ln(wth) = pexp ln 2 + ln(prod)
};

// A 4-entry group of the packed stream, unpacked: LDS byte offset of the minor's row and the count.
// The loop carries these (not the raw words) from one trip to the next: the unpacking is work
// the trip needs anyway, and a loop-carried value that is not a bare load result cannot be
// folded back into a load at the top of the consuming trip (which would undo the prefetch).
struct Group4 {
    uint32_t o0, o1, o2, o3;
    uint32_t c0, c1, c2, c3;
};
// Pins the unpacked group in VGPRs at this point of the trip: an empty asm the compiler can
// neither sink into the next trip nor look through.
__device__ __forceinline__ void pin(Group4 &g)
{
    asm volatile("" : "+v"(g.o0), "+v"(g.o1), "+v"(g.o2), "+v"(g.o3), "+v"(g.c0), "+v"(g.c1), "+v"(g.c2), "+v"(g.c3));
}
// the same for a trip that does not read the counts (the leading ones of a slice): only the row offsets are pinned
__device__ __forceinline__ void pin_offsets(Group4 &g)
{
    asm volatile("" : "+v"(g.o0), "+v"(g.o1), "+v"(g.o2), "+v"(g.o3));
}
// The entry stream is read exactly once per sweep (400 MB at the headline size, two sides): loaded with the non-temporal
// policy so that it does not push what IS re-read -- the factor rows, the per-task partial rows the update kernels gather
// back, the state -- out of the 4 MB L2s and the 256 MB Infinity Cache (MI355X_MICROARCH.md: a table stays resident only
// while everything moved between two uses of a line fits in about 256 MiB).  VBNMF_STREAM_NT=0 builds the plain loads (A/B).
#ifndef VBNMF_STREAM_NT
#define VBNMF_STREAM_NT 1
#endif
#ifndef VBNMF_PREFETCH2
#define VBNMF_PREFETCH2 0
#endif
#ifndef VBNMF_END_WB
#define VBNMF_END_WB 0
#endif
// nt (wave-uniform, SweepSide::stream_nt): the engine sets it when the step's streams do NOT fit the Infinity Cache.  Where they do
// (C2: 130 MB of entries; 5 000 x 20 000: 61 MB) the default policy keeps the stream itself on the die from one step to the next, and
// the non-temporal loads cost 1-3.5 % (profiles/r05_small_nt_ab.txt).
__device__ __forceinline__ uint4 ld_stream(const uint4 *p, int nt)
{
#if VBNMF_STREAM_NT
    if (nt) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
#endif
    return *p;
}
// The per-task partial rows (75 MB per sweep at the headline size) are written once and gathered back by the update kernels.
// VBNMF_PART_NT=1 stores them with the non-temporal policy (experiment: profiles/r05_part_nt_ab.txt).
#ifndef VBNMF_PART_NT
#define VBNMF_PART_NT 0
#endif
__device__ __forceinline__ void st_part(double2 *p, double2 v)
{
#if VBNMF_PART_NT
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<f64x2 *>(p));
#else
    *p = v;
#endif
}
// The update kernels gather every partial row once and write the means / variances (e, d) that nothing on the device reads
// again before the next update: VBNMF_UPD_NT=1 gives both the non-temporal policy (experiment: profiles/r05_upd_nt_ab.txt).
#ifndef VBNMF_UPD_NT
#define VBNMF_UPD_NT 0
#endif
__device__ __forceinline__ double ld_part(const double *p)
{
#if VBNMF_UPD_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_once(double *p, double v)
{
#if VBNMF_UPD_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ double2 ld_stream(const double2 *p, int nt)
{
#if VBNMF_STREAM_NT
    if (nt) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
        return make_double2(v.x, v.y);
    }
#endif
    return *p;
}
// LDS image of the sweep: [0, kLdsTabBytes) the ln table, then kLdsEvSlots per-slice evidence
// partials, the slice ticket counter, and from kLdsRowBase on the staged factor block.
constexpr uint32_t kLdsTabBytes = kLogTabSize * sizeof(LogTabEntry);
constexpr int kLdsEvSlots = 128;
constexpr uint32_t kLdsEvBase = kLdsTabBytes;
constexpr uint32_t kLdsCtrBase = kLdsEvBase + kLdsEvSlots * sizeof(double);
constexpr uint32_t kLdsRowBase = kLdsCtrBase + 16;
// One row (R doubles) of the staged factor block, by byte offset into the block.  The sweep kernels own no static
// LDS, so their dynamic LDS starts at address 0 (launch_sweep_t checks it) and the row address is the offset itself
// plus kLdsRowBase, a constant the ds_read instructions carry in their immediate: no VALU address arithmetic.
template <int R>
__device__ __forceinline__ void lds_row(const double2 *__restrict__ ldsG, uint32_t byte_off, double2 (&gv)[R / 2])
{
    (void)ldsG;
    typedef double __attribute__((ext_vector_type(2))) Pair;
    typedef const __attribute__((address_space(3))) Pair LdsPair;
    LdsPair *g = reinterpret_cast<LdsPair *>(static_cast<uintptr_t>(byte_off + kLdsRowBase));
#pragma unroll
    for (int kk = 0; kk < R / 2; kk++) { const Pair v = g[kk]; gv[kk] = make_double2(v.x, v.y); }
}

// share: byte offset of the lane's share of the row (ranks above 32, see sweep_side; 0 otherwise and folded away)
template <int R>
__device__ __forceinline__ Group4 unpack4(const uint4 e, uint32_t share = 0)
{
    // the word carries the row's byte offset inside the staged block (one AND) and the count (one shift), common.h
    Group4 g;
    g.o0 = (e.x & 0x3FFF0u) + share; g.o1 = (e.y & 0x3FFF0u) + share;
    g.o2 = (e.z & 0x3FFF0u) + share; g.o3 = (e.w & 0x3FFF0u) + share;
    g.c0 = e.x >> 18; g.c1 = e.y >> 18; g.c2 = e.z >> 18; g.c3 = e.w >> 18;
    return g;
}

// One stored entry: wth = F . g ; q = x / wth ; acc += q g ; lsum += x log(wth).
// Padding slots have x = 0 and minor 0 of the block: they add exact zeros as long as wth
// there is finite and non-zero.  It can only fail to be when a whole factor row is 0 or
// non-finite, and then the reference's dense X/wth (src/vbnmf_update.cpp:34) is NaN as well.
// SPMM: the plain product instead (acc += x g: no ratio, no logarithm) -- k_spmm, for the truncated SVD of the
// svd2 initialiser.
// ONE: the entry is a stored one (x is not read): q = 1 / wth, and on the side that carries the evidence's
// sum x log(wth) the logarithm is deferred -- the lane multiplies wth into a running product (renormalised every two
// entries, so any wth in 2^+-500 is safe) and takes ONE logarithm per slice: ~2.5 instructions per entry instead of ~18.
// Ranks above 32: SP = 2 or 4 neighbouring lanes share one task, each holding R of its R * SP columns (sweep_side);
// the dot product's shares are added across the SP lanes with DPP quad permutes (no LDS traffic).  a + b is
// commutative in IEEE arithmetic, so every lane of the group holds the same wth.
template <int SP>
__device__ __forceinline__ double share_sum(double v)
{
    if (SP >= 2) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, false));
    }
    if (SP >= 4) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, false));
    }
    return v;
}

// TWO: the entry is a stored one or two (the stretch behind the leading ones): the ordinary division, but the
// logarithm is deferred like the ones' -- wth or wth^2 goes into the running product, renormalised after every entry.
template <int R, bool SPMM = false, bool ONE = false, int SP = 1, bool TWO = false>
__device__ __forceinline__ void sweep_entry(SweepRegs<R> &S, const double2 *__restrict__ ldsG, const double2 (&gv)[R / 2],
                                            double x, bool logterm)
{
    if (ONE) {
        double w0 = S.F[0] * gv[0].x, w1 = S.F[1] * gv[0].y;
#pragma unroll
        for (int kk = 1; kk < R / 2; kk++) {
            w0 = fma(S.F[2 * kk], gv[kk].x, w0);
            w1 = fma(S.F[2 * kk + 1], gv[kk].y, w1);
        }
        const double wth = share_sum<SP>(w0 + w1);
        const double rc = sp_rcp_seed(wth);
        const double q = fma(fma(-wth, rc, 1.0), rc, rc);          // 1 / wth to 2^-48, as dev_div_fast
#pragma unroll
        for (int kk = 0; kk < R / 2; kk++) {
            S.acc[2 * kk] = fma(q, gv[kk].x, S.acc[2 * kk]);
            S.acc[2 * kk + 1] = fma(q, gv[kk].y, S.acc[2 * kk + 1]);
        }
        if (logterm) S.prod *= wth;
        return;
    }
    if (SPMM) {
#pragma unroll
        for (int kk = 0; kk < R / 2; kk++) {
            S.acc[2 * kk] = fma(x, gv[kk].x, S.acc[2 * kk]);
            S.acc[2 * kk + 1] = fma(x, gv[kk].y, S.acc[2 * kk + 1]);
        }
        return;
    }
    // two interleaved partial sums (even / odd k): half the dependent-chain length
    double w0 = S.F[0] * gv[0].x, w1 = S.F[1] * gv[0].y;
#pragma unroll
    for (int kk = 1; kk < R / 2; kk++) {
        w0 = fma(S.F[2 * kk], gv[kk].x, w0);
        w1 = fma(S.F[2 * kk + 1], gv[kk].y, w1);
    }
    const double wth = share_sum<SP>(w0 + w1);
    const double q = dev_div_fast(x, wth);
#pragma unroll
    for (int kk = 0; kk < R / 2; kk++) {
        S.acc[2 * kk] = fma(q, gv[kk].x, S.acc[2 * kk]);
        S.acc[2 * kk + 1] = fma(q, gv[kk].y, S.acc[2 * kk + 1]);
    }
    if (TWO) {
        if (logterm) S.prod *= (x == 2.0) ? wth * wth : wth;
        return;
    }
    if (logterm) S.lsum = fma(x, dev_log_tab(wth, reinterpret_cast<const LogTabEntry *>(ldsG)), S.lsum);
}

template <int R>
__device__ __forceinline__ void renorm_product(SweepRegs<R> &S)
{
    int k;
    S.prod = sp_frexp(S.prod, &k);
    S.pexp += k;
}

// Column sums over the 64 lanes of a wave of N values per lane, by recursive halving: at every level the lanes pair up, the
// pair splits the columns -- one lane keeps the lower half, its partner the upper half, each adding the other's copy -- so
// about N + log2(64) exchanges instead of 6 N; once one value per lane is left the remaining levels are plain pair sums
// (one lane of the pair keeps the result).  Levels (bit of the lane number that tells the two halves apart : pairing):
//   32 : lane ^ 32   v_permlane32_swap (gfx950: swaps the upper 32 lanes of one register with the lower 32 of another, so
//   16 : lane ^ 16   v_permlane16_swap  `lo' = (lo | hi swapped); w = lo' + hi'` needs no select: 3 instructions per column)
//    8 : lane ^ 8    DPP row_ror:8
//    4 : 7 - lane within its 8 lanes   DPP row_half_mirror (any pairing of a low-quad lane with a high-quad lane will do;
//        lane ^ 4 through ds_bpermute when LOW > 1: the pairing must then keep the lane's share of the task's columns)
//    2 : lane ^ 2    DPP quad_perm [2,3,0,1]
//    1 : lane ^ 1    DPP quad_perm [1,0,3,2]
// Levels below LOW are not crossed (ranks above 32: the LOW = SP neighbouring lanes hold different columns of one task).
// On return the lane holds `onv` (0, 1 or 2) finished column sums o0, o1 of columns obase, obase + 1.  Which lane ends
// with which column depends on the lane number alone, and the tree is fixed: a slice's sums depend on nothing but its
// tasks -- bit-reproducible.
// XOR4: level 4 must pair lane ^ 4 (ranks above 32: the mirror pairing joins lanes that hold different columns of a task)
template <int MASK, bool XOR4 = false>
__device__ __forceinline__ double lane_pair_value(double v)
{
    static_assert(MASK == 8 || MASK == 4 || MASK == 2 || MASK == 1, "DPP levels");
    if constexpr (MASK == 4 && XOR4) return __shfl_xor(v, 4, 64);
    constexpr int ctrl = MASK == 8 ? 0x128 : MASK == 4 ? 0x141 : MASK == 2 ? 0x4E : 0xB1;
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, false),
                            __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, false));
}
// wave sum without LDS round trips: DPP inside the rows of 16 lanes (quads, half rows, rows), then gfx950's permlane swaps
// across the rows.  Valid in every lane; a fixed tree.
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    v += lane_pair_value<1>(v);
    v += lane_pair_value<2>(v);
    v += lane_pair_value<4>(v);
    v += lane_pair_value<8>(v);
    {   // rows 0+1, 2+3: v_permlane16_swap of the value with a copy of itself leaves (own row, partner row) in the two registers
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
    }
    {   // the two halves of the wave
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
    }
    return v;
}

// lower lanes of the level: lo + partner's lo ; upper lanes: hi + partner's hi
template <int MASK, bool XOR4 = false>
__device__ __forceinline__ double lane_pair_split(double lo, double hi, bool up)
{
    if constexpr (MASK == 32 || MASK == 16) {
        (void)up;
        int a0 = __double2loint(lo), a1 = __double2hiint(lo), b0 = __double2loint(hi), b1 = __double2hiint(hi);
        if constexpr (MASK == 32) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        } else {
            const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        }
    } else {
        const double send = up ? lo : hi, keep = up ? hi : lo;
        return keep + lane_pair_value<MASK, XOR4>(send);
    }
}
template <int N, int MASK, int LOW>
__device__ __forceinline__ void colsum_fold(const double (&v)[N], int lane, int base, int nv, double &o0, double &o1, int &obase, int &onv)
{
    if constexpr (MASK < LOW) {
        static_assert(N <= 2, "colsum_fold: more than two columns left per lane");
        o0 = v[0]; o1 = (N > 1) ? v[N > 1 ? 1 : 0] : 0.0; obase = base; onv = nv;
    } else if constexpr (N == 1) {
        double w[1];
        if constexpr (MASK >= 16) w[0] = v[0] + __shfl_xor(v[0], MASK, 64);
        else w[0] = v[0] + lane_pair_value<MASK, (LOW > 1)>(v[0]);
        colsum_fold<1, MASK / 2, LOW>(w, lane, base, (lane & MASK) ? 0 : nv, o0, o1, obase, onv);
    } else {
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & MASK) != 0;
        double w[H];
#pragma unroll
        for (int i = 0; i < H; i++) w[i] = lane_pair_split<MASK, (LOW > 1)>(v[i], (i + H < N) ? v[(i + H < N) ? i + H : 0] : 0.0, up);
        colsum_fold<H, MASK / 2, LOW>(w, lane, base + (up ? H : 0), up ? max(nv - H, 0) : min(nv, H), o0, o1, obase, onv);
    }
}

// Next slice ticket of the workgroup, wave-uniform by construction: EVERY lane adds 1 to the LDS counter (the
// compiler folds the 64 adds into one ds_add_rtn of 64 by the first active lane), so there is no divergent branch
// in the source, the old value is a multiple of 64 whichever lane's return is read, and the ticket old / 64 lives in
// an SGPR: the slice loop's exit is a scalar compare-and-branch (s_cmp_ge_i32 / s_cbranch_scc1 in the ISA).
// Why not "lane 0 adds, the others take i = 0, then __shfl / readfirstlane": inlined into the slice loop, hipcc
// (ROCm 7.2) threads the back edge of the lanes that skip the atomic -- for which i = 0 is a compile-time constant --
// straight to the exit test, i.e. it splits the loop into an outer one (lane 0: ds_add_rtn) and an inner one that the
// other 63 lanes re-enter with v6 = 0 while lane 0 has left it; their ds_bpermute / v_readfirstlane then reads an
// inactive (or the wrong first active) lane, gets ticket 0 < cn again and the wave never leaves slice 0.  (Seen in
// the ISA of k_sweep<10, false, 768>: loop header `ds_bpermute_b32 v108, v98, v6`, latch `v_mov_b32 v6, 0` behind
// an exec mask built from the lane == 0 predicate; the same nest appears with readfirstlane.)  LLVM's convergent
// attribute does not pin which lanes meet at a cross-lane operation inside a loop, so the rewrite is not a bug it
// acknowledges; round 1 hid it behind __attribute__((noinline)).  Without a divergent branch there is nothing to thread.
__device__ __forceinline__ int take_ticket(int *ticket)
{
    const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(old) >> 6;
}

// The same list pulled from BOTH ends: the word's low half counts the slices taken from the front (longest first),
// its high half those taken from the back (shortest first); a pull is valid while front + back < cn and then names
// front (or cn - 1 - back): every slice exactly once.  The youngest waves of the workgroup pull from the back: the SIMD
// issues its oldest wave first, so a young wave crawls through whatever it holds -- holding a LONG slice when the
// tickets run out is what leaves one wave per SIMD running alone at the end of a side (waves 0-3 / 4-7 / 8-11 left the
// gene side at 77 / 85 / 93 us).  Same construction as above: every lane adds the (wave-uniform) increment.
__device__ __forceinline__ int take_ticket_ends(int *ticket, int from_back, int cn)
{
    const int inc = __builtin_amdgcn_readfirstlane(from_back ? (1 << 16) : 1);
    const int old = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(ticket, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    const int front = (old & 0xFFFF) >> 6, back = (int)((unsigned)old >> 22);
    if (front + back >= cn) return cn;
    // (the ticket must be an SGPR for the compiler too: with a lane-valued ticket the slice loop's exit is a divergent
    // branch and the slice header loads become per-lane -- 150 more wave instructions per slice, measured)
    return __builtin_amdgcn_readfirstlane(from_back ? cn - 1 - back : front);
}

// EV selects the per-slice scalar left in the evidence slots: 1 = the VB data term sum(acc . llF) - sum x log(wth)
// (see the header), 2 = sum x log(wth) alone (ML-NMF likelihood, mlnmf.h), 0 = nothing, 3 = nothing and the entries
// are accumulated as a plain sparse product (k_spmm).
// SP (1, 2 or 4): ranks above 32.  The factor rows have RT = R * SP columns; SP neighbouring lanes share a task, lane
// share hp holding columns [hp R, (hp + 1) R) of the major's row, of the accumulators and of every gathered row.  A
// slice still has 64 tasks: the wave runs it as SP sub-slices of 64 / SP tasks, one after the other.
// The workgroup's sums of its slices' column-sum rows (SweepSide::csl, written by the gene side of the VB sweep), in list
// order: one wave, lane c owns columns c and c + 64.  The workgroup's slices are a contiguous id range (slices are
// numbered in processing order), so are their rows.  Called once the rows are complete and visible: behind a
// __syncthreads() that follows the side that wrote them (workgroup scope: the same CU stored them).
template <int RT>
__device__ __forceinline__ void colsum_finish(const SweepSide &A, int wg, int lane)
{
    const int s0 = A.seg_ptr[A.wg_seg0[wg]], s1 = A.seg_ptr[A.wg_seg0[wg + 1]];
    const int cn = s1 - s0;
    const double *rows = A.csl + (size_t)s0 * RT;
    for (int col = lane; col < RT; col += 64) {
        double t = 0.0;
        for (int q = 0; q < cn; q += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = rows[(size_t)min(q + u, cn - 1) * RT + col];
#pragma unroll
            for (int u = 0; u < 16; u++) t += (q + u < cn) ? v[u] : 0.0;
        }
        A.csum[(size_t)wg * RT + col] = t;                 // (a workgroup without slices leaves 0: the update adds all n_wg rows)
    }
}

template <int R, bool WIDE, bool LOGTERM, int NT, int EV = 1, int SP = 1>
__device__ __forceinline__ void sweep_side(const SweepSide &S, double2 *__restrict__ ldsG, const SweepSide *job = nullptr)
{
    constexpr int RT = R * SP;
    constexpr int TL = 64 / SP;                            // tasks per sub-slice
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tl = lane / SP, hp = lane % SP;              // task lane inside the sub-slice, share of the row
    const uint32_t share = (uint32_t)hp * (R * 8);
    // workgroups that share an XCD (blockIdx % 8) take neighbouring ranges, i.e. mostly the same blocks
    const int nwg = S.n_wg;
    const int wg = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int seg0 = S.wg_seg0[wg], seg1 = S.wg_seg0[wg + 1];
    if (S.dbg && lane == 0) S.dbg[(size_t)wg * (2 + 2 * (NT / 64)) + 2 + 2 * wave] = __builtin_amdgcn_s_memrealtime();
    if (S.dbg && threadIdx.x == 0) S.dbg[(size_t)wg * (2 + 2 * (NT / 64))] = __builtin_amdgcn_s_memrealtime();
    double ev_wg = 0.0;                                    // this workgroup's evidence, summed in list order (thread 0)
    // CS: the gene side of the VB sweep also leaves sum_i sw_ik = sum over tasks of F_k acc_k (S.csl / S.csum), which lets
    // ONE launch run both posterior updates (k_update2): the H update's rate needs colSums(ew_new) (reference
    // src/vbnmf_update.cpp:53), and sum_i ew_ik = (n aw + sum_i sw_ik) / bew_k is known without the W update having run.
    constexpr bool CS = (EV == 1) && LOGTERM;
    const bool cs_on = CS && S.csl != nullptr;             // (workgroup-uniform)
    const int snt = S.stream_nt;                           // the entry stream's cache policy (ld_stream)
    // The per-slice rows are added up per workgroup by colsum_finish(): NOT here -- their read-back is a global round trip,
    // and inside this side's chunk ends it held the whole workgroup for 3 us at rank 10 and 14 us at rank 20 (round 5,
    // profiles/r05_pair_ab.txt).  k_sweep hands the job to the CELL side, whose wave 1 runs it when it has run out of
    // slices, in front of the barrier that waits for the slower waves anyway.
    bool job_pending = job != nullptr && job->csl != nullptr;   // (workgroup-uniform)
    const int from_back = __builtin_amdgcn_readfirstlane((int)(wave >= (NT / 64) - S.pull_ends));   // the youngest waves (take_ticket_ends)
    double *ev_slot = reinterpret_cast<double *>(reinterpret_cast<char *>(ldsG) + kLdsEvBase);
    int *ticket = reinterpret_cast<int *>(reinterpret_cast<char *>(ldsG) + kLdsCtrBase);
    for (int seg = seg0; seg < seg1; seg++) {
        const int blk = S.seg_block[seg];
        const int m0 = S.block_start[blk];
        const int cw = S.block_start[blk + 1] - m0;
        __syncthreads();                                   // readers of the previous block are done
        if (threadIdx.x == 0) *ticket = 0;
        {
            const int kSlots = S.row_slots;                // LDS row stride in 16-byte slots (odd)
            const double2 *G2 = reinterpret_cast<const double2 *>(S.G + (size_t)m0 * RT);
            const int cnt = cw * (RT / 2);
            double2 *rows = ldsG + kLdsRowBase / sizeof(double2);
            if (kSlots == RT / 2) {                        // rows as long as their stride (R / 2 odd, own geometry): a straight copy
                for (int t = threadIdx.x; t < cnt; t += NT) rows[t] = G2[t];
            } else {
                const int pad = kSlots - RT / 2;
                for (int t = threadIdx.x; t < cnt; t += NT) rows[t + (t / (RT / 2)) * pad] = G2[t];
            }
            if (threadIdx.x < kLogTabSize) ldsG[threadIdx.x] = reinterpret_cast<const double2 *>(S.logtab)[threadIdx.x];
        }
        __syncthreads();
        // The waves pull slices from the segment's list (longest first) through an LDS ticket counter, in
        // chunks of kLdsEvSlots so every slice's evidence partial has an LDS slot; after a chunk the slots
        // are added in list order, so the sum does not depend on which wave ran which slice.
        const int l0 = S.seg_ptr[seg], l1 = S.seg_ptr[seg + 1];
        for (int c0 = l0; c0 < l1; c0 += kLdsEvSlots) {
        const int cn = min(kLdsEvSlots, l1 - c0);
        while (true) {
            const int i = S.pull_ends ? take_ticket_ends(ticket, from_back, cn) : take_ticket(ticket);
            if (i >= cn) break;
            const int s = c0 + i;                          // slices are numbered in processing order
            const int ng = S.slice_width[s] >> 2;
            const int64_t off = S.slice_off[s];
            double ev = 0.0;                               // the lane's evidence contribution over the sub-slices
            double cs0 = 0.0, cs1 = 0.0;                   // (CS) this lane's finished column sums of the slice ...
            int cs_base = 0, cs_nv = 0;                    // ... their first column and how many of them it holds
#pragma unroll 1
            for (int sub = 0; sub < SP; sub++) {
            const int tlane = sub * TL + tl;               // the task's lane in the slice's lane-interleaved storage
            const uint32_t M = S.task_major[(size_t)s * 64 + tlane];
            SweepRegs<R> T;
            if (M != kIdle) {
                const double2 *F2 = reinterpret_cast<const double2 *>(S.F + (size_t)M * RT + hp * R);
#pragma unroll
                for (int kk = 0; kk < R / 2; kk++) { double2 v = F2[kk]; T.F[2 * kk] = v.x; T.F[2 * kk + 1] = v.y; }
            } else {
#pragma unroll
                for (int k = 0; k < R; k++) T.F[k] = 1.0;   // idle lane: only sees padding slots
            }
#pragma unroll
            for (int k = 0; k < R; k++) T.acc[k] = 0.0;
            T.lsum = 0.0;
            T.prod = 1.0;
            T.pexp = 0;

            // Slice widths are multiples of 4 (one group of the packed stream): two 4-entry groups per trip, the loads
            // of the next trip issued before this trip's arithmetic, LDS rows fetched one entry ahead; an odd last
            // group runs as half a trip behind the loop.
            const int np = ng >> 1;
            double2 g0[R / 2];
            if ((R >= VBNMF_ONEBUF_FROM || R <= VBNMF_ONEBUF_UPTO) && !WIDE) {
                // very large ranks: the factor row, the accumulators and ONE gathered row already fill
                // the register file, so no second row buffer and no look-ahead here
                const uint4 *E = reinterpret_cast<const uint4 *>(S.packed + off) + tlane;
                const int ngf = (EV == 3) ? 0 : min(ng, (S.slice_fast[s] & 0xFFFF) >> 2);      // groups inside the leading stretch of ones
                int g = 0;
                for (; g < ngf; g++) {
                    const Group4 a = unpack4<R>(ld_stream(E + (size_t)g * 64, snt), share);
                    lds_row<R>(ldsG, a.o0, g0); sweep_entry<R, false, true, SP>(T, ldsG, g0, 1.0, LOGTERM);
                    lds_row<R>(ldsG, a.o1, g0); sweep_entry<R, false, true, SP>(T, ldsG, g0, 1.0, LOGTERM);
                    if (LOGTERM) renorm_product<R>(T);
                    lds_row<R>(ldsG, a.o2, g0); sweep_entry<R, false, true, SP>(T, ldsG, g0, 1.0, LOGTERM);
                    lds_row<R>(ldsG, a.o3, g0); sweep_entry<R, false, true, SP>(T, ldsG, g0, 1.0, LOGTERM);
                    if (LOGTERM) renorm_product<R>(T);
                }
                for (; g < ng; g++) {
                    const Group4 a = unpack4<R>(ld_stream(E + (size_t)g * 64, snt), share);
                    lds_row<R>(ldsG, a.o0, g0); sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, (double)a.c0, LOGTERM);
                    lds_row<R>(ldsG, a.o1, g0); sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, (double)a.c1, LOGTERM);
                    lds_row<R>(ldsG, a.o2, g0); sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, (double)a.c2, LOGTERM);
                    lds_row<R>(ldsG, a.o3, g0); sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, (double)a.c3, LOGTERM);
                }
                if (LOGTERM && ngf > 0)
                    T.lsum += fma((double)T.pexp, 6.93147180559945286227e-01, dev_log_tab(T.prod, reinterpret_cast<const LogTabEntry *>(ldsG)));
            } else if (!WIDE) {
                double2 g1[R / 2];
                const uint4 *E = reinterpret_cast<const uint4 *>(S.packed + off) + tlane;
                Group4 a = unpack4<R>(ld_stream(E, snt), share), b = unpack4<R>(ld_stream(E + (size_t)min(1, ng - 1) * 64, snt), share);
#if VBNMF_PREFETCH2
                // the entry stream TWO trips ahead: the groups of trip p + 1 are on their way (ec, ed) when trip p issues the
                // loads of trip p + 2 -- with the stream read non-temporally every group comes from HBM, 1.5-2 us under load,
                // about what ONE trip of three interleaved waves takes
                uint4 ec = ld_stream(E + (size_t)min(2, ng - 1) * 64, snt), ed = ld_stream(E + (size_t)min(3, ng - 1) * 64, snt);
#endif
                lds_row<R>(ldsG, a.o0, g0);
                // FENCE keeps the machine scheduler from sinking a row's LDS reads down to their first
                // use: the reads of entry j+1 stay in front of the arithmetic of entry j, which hides them.
#define VBNMF_FENCE() __builtin_amdgcn_sched_barrier(0)
                // One trip = 8 entries (two 4-entry groups).  ONE = the trip lies in the slice's leading stretch of
                // stored ones (layout: slice_fast), PIN = the matching pin of the next trip's unpacked groups.
                // MODE: 0 general, 1 the leading ones, 2 the ones-or-twos behind them (gene side only: it defers the logarithm)
#define VBNMF_ENTRY(MODE, gv, cnt) sweep_entry<R, EV == 3, (MODE) == 1, SP, (MODE) == 2>(T, ldsG, gv, (double)(cnt), LOGTERM); if ((MODE) == 2 && LOGTERM) renorm_product<R>(T); VBNMF_FENCE()
#if VBNMF_PREFETCH2
#define VBNMF_TRIP_LOADS const uint4 fc = ld_stream(E + (size_t)min(2 * p + 4, ng - 1) * 64, snt), fd = ld_stream(E + (size_t)min(2 * p + 5, ng - 1) * 64, snt);
#define VBNMF_TRIP_ROTATE ec = fc; ed = fd;
#else
#define VBNMF_TRIP_LOADS const uint4 ec = ld_stream(E + (size_t)min(2 * p + 2, ng - 1) * 64, snt), ed = ld_stream(E + (size_t)min(2 * p + 3, ng - 1) * 64, snt);
#define VBNMF_TRIP_ROTATE
#endif
#define VBNMF_TRIP(MODE, PIN)                                                                     \
                {                                                                                 \
                    /* the next trip's groups; past the end: the last group again (an odd one is the tail's) */ \
                    VBNMF_TRIP_LOADS                                                              \
                    lds_row<R>(ldsG, a.o1, g1); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g0, a.c0);                                                   \
                    lds_row<R>(ldsG, a.o2, g0); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g1, a.c1);                                                   \
                    if ((MODE) == 1 && LOGTERM) renorm_product<R>(T);                                     \
                    lds_row<R>(ldsG, a.o3, g1); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g0, a.c2);                                                   \
                    lds_row<R>(ldsG, b.o0, g0); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g1, a.c3);                                                   \
                    if ((MODE) == 1 && LOGTERM) renorm_product<R>(T);                                     \
                    lds_row<R>(ldsG, b.o1, g1); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g0, b.c0);                                                   \
                    lds_row<R>(ldsG, b.o2, g0); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g1, b.c1);                                                   \
                    if ((MODE) == 1 && LOGTERM) renorm_product<R>(T);                                     \
                    lds_row<R>(ldsG, b.o3, g1); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g0, b.c2);                                                   \
                    a = unpack4<R>(ec, share);                                                    \
                    PIN(a);                                                                       \
                    lds_row<R>(ldsG, a.o0, g0); VBNMF_FENCE();                                    \
                    VBNMF_ENTRY(MODE, g1, b.c3);                                                   \
                    if ((MODE) == 1 && LOGTERM) renorm_product<R>(T);                                     \
                    b = unpack4<R>(ed, share);                                                    \
                    PIN(b);                                                                       \
                    VBNMF_TRIP_ROTATE                                                             \
                }
                const int sfast = S.slice_fast[s];
                const int npf = (EV == 3) ? 0 : min(np, (sfast & 0xFFFF) >> 3);
                const int npf2 = (EV == 3 || !LOGTERM) ? npf : min(np, sfast >> 19);    // trips of ones or twos (>= npf)
                int p = 0;
                for (; p < npf - 1; p++) VBNMF_TRIP(1, pin_offsets)
                if (p < npf) { VBNMF_TRIP(1, pin) p++; }                  // the next trip reads the counts again
                for (; p < npf2; p++) VBNMF_TRIP(2, pin)
                for (; p < np; p++) VBNMF_TRIP(0, pin)
                if (ng & 1) {                                             // the odd last group: `a` holds it, its first row is in g0
                    lds_row<R>(ldsG, a.o1, g1); VBNMF_FENCE();
                    VBNMF_ENTRY(0, g0, a.c0);
                    lds_row<R>(ldsG, a.o2, g0); VBNMF_FENCE();
                    VBNMF_ENTRY(0, g1, a.c1);
                    lds_row<R>(ldsG, a.o3, g1); VBNMF_FENCE();
                    VBNMF_ENTRY(0, g0, a.c2);
                    VBNMF_ENTRY(0, g1, a.c3);
                }
                if (LOGTERM && npf2 > 0)                                  // the deferred logarithm of the leading ones and twos
                    T.lsum += fma((double)T.pexp, 6.93147180559945286227e-01, dev_log_tab(T.prod, reinterpret_cast<const LogTabEntry *>(ldsG)));
#undef VBNMF_TRIP
#undef VBNMF_TRIP_LOADS
#undef VBNMF_TRIP_ROTATE
#undef VBNMF_ENTRY
#undef VBNMF_FENCE
            } else {
                double2 g1[R / 2];
                const uint4 *E = reinterpret_cast<const uint4 *>(S.widx + off) + tlane;
                const double2 *V = reinterpret_cast<const double2 *>(S.wval + off) + tlane * 2;
                const uint32_t rowb = (uint32_t)S.row_slots * 16u;     // bytes per staged row
                for (int g = 0; g < ng; g++) {
                    const uint4 c = ld_stream(E + (size_t)g * 64, snt);
                    const double2 v0 = ld_stream(V + (size_t)g * 128, snt), v1 = ld_stream(V + (size_t)g * 128 + 1, snt);
                    lds_row<R>(ldsG, c.x * rowb + share, g0);
                    lds_row<R>(ldsG, c.y * rowb + share, g1);
                    sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, v0.x, LOGTERM);
                    lds_row<R>(ldsG, c.z * rowb + share, g0);
                    sweep_entry<R, EV == 3, false, SP>(T, ldsG, g1, v0.y, LOGTERM);
                    lds_row<R>(ldsG, c.w * rowb + share, g1);
                    sweep_entry<R, EV == 3, false, SP>(T, ldsG, g0, v1.x, LOGTERM);
                    sweep_entry<R, EV == 3, false, SP>(T, ldsG, g1, v1.y, LOGTERM);
                }
            }

            // partial statistics of this task and the lane's evidence contribution
            {
                double2 *P = reinterpret_cast<double2 *>(S.part + ((size_t)s * 64 + tlane) * RT + hp * R);
#pragma unroll
                for (int kk = 0; kk < R / 2; kk++) st_part(P + kk, make_double2(T.acc[2 * kk], T.acc[2 * kk + 1]));
            }
            if (EV == 1 && M != kIdle) {
                const double2 *L2 = reinterpret_cast<const double2 *>(S.llF + (size_t)M * RT + hp * R);
                double evt = 0.0;
#pragma unroll
                for (int kk = 0; kk < R / 2; kk++) {
                    const double2 l = L2[kk];
                    evt = fma(T.acc[2 * kk], l.x, evt);
                    evt = fma(T.acc[2 * kk + 1], l.y, evt);
                }
                if (hp == 0) evt -= T.lsum;               // every lane of the group holds the task's sum x log(wth)
                ev += evt;
            }
            if (EV == 2 && M != kIdle && hp == 0) ev += T.lsum;
            if (CS && cs_on) {                             // column sums of F .* acc over the sub-slice's tasks
                double c[R];
#pragma unroll
                for (int k = 0; k < R; k++) c[k] = (M != kIdle) ? T.F[k] * T.acc[k] : 0.0;
                double o0, o1;
                colsum_fold<R, 32, SP>(c, lane, 0, R, o0, o1, cs_base, cs_nv);
                cs0 += o0; cs1 += o1;
            }
            }                                              // sub-slices
            ev = wave_sum_dpp(ev);                         // (every lane holds the slice's evidence partial)
            if (lane == 0) ev_slot[i] = ev;
            if (CS && cs_on) {                             // ... and its row of column sums (added up by colsum_finish)
                double *row = S.csl + (size_t)s * RT + hp * R + cs_base;
                if (cs_nv >= 1) row[0] = cs0;
                if (cs_nv >= 2) row[1] = cs1;
            }
        }
        // end of the chunk: add its slots in list order (wave 0: lane-strided, then a fixed shuffle tree)
        if (job_pending) {                                 // the other side's column sums, by a wave that is out of slices
            if (wave == 1) colsum_finish<RT>(*job, wg, lane);
            job_pending = false;
        }
        __syncthreads();
        if (wave == 0) {
            double t = 0.0;
            for (int q = lane; q < cn; q += 64) t += ev_slot[q];
            ev_wg += wave_sum_dpp(t);
            if (lane == 0) *ticket = 0;
        }
        __syncthreads();
        }
    }
    if (job_pending && wave == 1) colsum_finish<RT>(*job, wg, lane);      // (no chunk end on this side: a workgroup without slices)
    if (S.dbg && lane == 0) S.dbg[(size_t)wg * (2 + 2 * (NT / 64)) + 3 + 2 * wave] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        S.epart[wg] = ev_wg;                               // one evidence partial per workgroup and side
        if (S.dbg) S.dbg[(size_t)wg * (2 + 2 * (NT / 64)) + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int R, bool WIDE, int NT, int SP = 1>
__global__ __launch_bounds__(NT) void k_sweep(const SweepSide A, const SweepSide B)
{
    extern __shared__ double2 ldsG[];
    if (A.stop && *A.stop) return;               // the driver loop has ended: leave the statistics as they are
    sweep_side<R, WIDE, true, NT, 1, SP>(A, ldsG);      // lanes own genes: statistics sw + the sum x log(wth) (+ rows of column sums)
    sweep_side<R, WIDE, false, NT, 1, SP>(B, ldsG, &A); // lanes own cells: statistics sh (+ the gene side's column sums added up)
#if VBNMF_END_WB
    // experiment (profiles/r05_end_wb_ab.txt): every workgroup asks its XCD's L2 to write back when IT is done, so that the
    // release at the kernel's end -- the 6.5 us between this kernel and the next -- finds little left to write
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}

// The sweeps of a batch of engines (k_update2_batch above): jobs[2 b] = gene side, jobs[2 b + 1] = cell side of engine b.
template <int R, bool WIDE, int NT, int SP = 1>
__global__ __launch_bounds__(NT) void k_sweep_batch(const SweepSide *__restrict__ jobs)
{
    extern __shared__ double2 ldsG[];
    const SweepSide A = jobs[2 * blockIdx.y], B = jobs[2 * blockIdx.y + 1];
    if (A.stop && *A.stop) return;
    sweep_side<R, WIDE, true, NT, 1, SP>(A, ldsG);
    sweep_side<R, WIDE, false, NT, 1, SP>(B, ldsG, &A);
}

// One side alone (ML-NMF: its H and W updates are sequential, reference R/factorize.R:8-24, so each needs its
// own pass over X; the cell-side pass also yields the likelihood's sum x log(wh)).
// VB = true: one side of the VB sweep launched alone, with the VB evidence partial (EV = 1): cell-partitioned engines
// run the gene side first, so that its statistics can be all-reduced while the cell side runs.
template <int R, bool WIDE, bool LOGTERM, int NT, bool VB = false, int SP = 1>
__global__ __launch_bounds__(NT) void k_sweep1(const SweepSide S)
{
    extern __shared__ double2 ldsG[];
    if (S.stop && *S.stop) return;               // device-driven loop: the run has ended
    sweep_side<R, WIDE, LOGTERM, NT, VB ? 1 : (LOGTERM ? 2 : 0), SP>(S, ldsG);
}

// Sparse product on the tiled layout (the truncated SVD behind the svd2 initialiser,
// reference R/bayesian.R:150-159 irlba): per task sum_minor x * G[minor, :] -- X G on the gene side, t(X) G on the
// cell side -- into the same per-task partials, which k_pack then sums per major.
template <int R, bool WIDE, int NT, int SP = 1>
__global__ __launch_bounds__(NT) void k_spmm(const SweepSide S)
{
    extern __shared__ double2 ldsG[];
    sweep_side<R, WIDE, false, NT, 3, SP>(S, ldsG);
}

// Sum of one major's task partials for column k, in the inverse index's fixed order.  The
// task ids and then the partials are loaded VBNMF_GATHER_WIDTH at a time so the loads overlap.
// 16 since round 3 (8 before): 200 x 500 at rank 3 31.2 -> 30.3 us per step, 2 000 x 10 000 at rank 5 78.5 -> 77.7, the
// headline unchanged (its gather is bound by the sectors it moves, not by round trips); 121 VGPRs, no spill.
#ifndef VBNMF_GATHER_WIDTH
#define VBNMF_GATHER_WIDTH 16
#endif
__device__ __forceinline__ double task_sum(const double *__restrict__ part, const uint32_t *__restrict__ inv_task,
                                           int q0, int q1, int R, int k)
{
    double s = 0.0;
    constexpr int NF = VBNMF_GATHER_WIDTH;
    for (int q = q0; q < q1; q += NF) {
        uint32_t id[NF];
        double v[NF];
#pragma unroll
        for (int u = 0; u < NF; u++) id[u] = inv_task[min(q + u, q1 - 1)];
#pragma unroll
        for (int u = 0; u < NF; u++) v[u] = ld_part(part + (size_t)id[u] * R + k);
#pragma unroll
        for (int u = 0; u < NF; u++) s += (q + u < q1) ? v[u] : 0.0;
    }
    return s;
}

// The same with the ids read from an LDS copy of the block's stretch of the inverse index (k_update stages it once per
// block): the value loads of a major no longer wait for a global round trip that fetches their ids, and the rounds of one
// sum are independent of each other.  Same summation order: bit-identical results.
__device__ __forceinline__ double task_sum_lds(const double *__restrict__ part, const uint32_t *ids, int q0, int q1, int R, int k)
{
    double s = 0.0;
    constexpr int NF = VBNMF_GATHER_WIDTH;
    for (int q = q0; q < q1; q += NF) {
        double v[NF];
#pragma unroll
        for (int u = 0; u < NF; u++) v[u] = ld_part(part + (size_t)ids[min(q + u, q1 - 1)] * R + k);
#pragma unroll
        for (int u = 0; u < NF; u++) s += (q + u < q1) ? v[u] : 0.0;
    }
    return s;
}

// ------------------------------------------------------------------------------------
// Small fixed-order reductions used by the update / final kernels.
// ------------------------------------------------------------------------------------
constexpr int kUpdateBlocks = 256;     // persistent blocks of k_update / k_prime (one per CU)
constexpr int kUpdateThreads = 1024;
constexpr int kStageIds = 14336;       // task ids of a block's majors staged in LDS by k_update (56 KB) ...
constexpr int kStagePtr = 2048;        // ... and their pointer stretch (8 KB); a block with more falls back to global reads

__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }     // (valid in every lane; no LDS round trips)

// Column sums of a block-partials table bp[nb][ncol] (nb <= 256) into out[0..ncol) (LDS or
// global): wave w takes columns w, w+nwaves, ...; lane l adds rows l, l+64, ... in order.
__device__ __forceinline__ void bp_colsums(const double *__restrict__ bp, int nb, int ncol, double *out, int nthreads)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthreads >> 6;
    for (int c = wave; c < ncol; c += nw) {
        double s = 0.0;
        for (int b = lane; b < nb; b += 64) s += bp[(size_t)b * ncol + c];
        s = wave_sum(s);
        if (lane == 0) out[c] = s;
    }
}

// The same for two tables at once: the loads of both are in flight together (one memory latency instead of two).
__device__ __forceinline__ void bp_colsums2(const double *__restrict__ bpA, const double *__restrict__ bpB, int nb, int ncol,
                                            double *outA, double *outB, int nthreads)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthreads >> 6;
    for (int c = wave; c < ncol; c += nw) {
        double sa = 0.0, sb = 0.0;
        for (int b = lane; b < nb; b += 64) { sa += bpA[(size_t)b * ncol + c]; sb += bpB[(size_t)b * ncol + c]; }
        sa = wave_sum(sa); sb = wave_sum(sb);
        if (lane == 0) { outA[c] = sa; outB[c] = sb; }
    }
}

// Three tables at once (k_update2's prologue): A and B have ncol columns, C has ncolC <= ncol; nbC rows for C.
__device__ __forceinline__ void bp_colsums3(const double *__restrict__ bpA, const double *__restrict__ bpB, int nb, int ncol,
                                            const double *__restrict__ bpC, int nbC, int ncolC,
                                            double *outA, double *outB, double *outC, int nthreads)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthreads >> 6;
    for (int c = wave; c < ncol; c += nw) {
        double sa = 0.0, sb = 0.0, sc = 0.0;
        for (int b = lane; b < nb; b += 64) { if (bpA) sa += bpA[(size_t)b * ncol + c]; sb += bpB[(size_t)b * ncol + c]; }
        if (c < ncolC) for (int b = lane; b < nbC; b += 64) sc += bpC[(size_t)b * ncolC + c];
        sa = wave_sum_dpp(sa); sb = wave_sum_dpp(sb); sc = wave_sum_dpp(sc);
        if (lane == 0) { if (bpA) outA[c] = sa; outB[c] = sb; if (c < ncolC) outC[c] = sc; }
    }
}

// ------------------------------------------------------------------------------------
// Posterior update of one factor (both sides share it).  256 persistent blocks; thread
// (row_sub, k) walks its block's majors with a fixed k.
//   s   = acc[major][k] (dense), or the sum of the major's task partials (inv_ptr != null)
//   al  = a + l_old * s                     :38-39 / :48-49
//   be  = a/b + other[k]                    :40-43 (rowSums of the incoming eh) / :50-53 (colSums of the NEW ew)
//   e   = al/be ; d = al/be/be              :44,46 / :54,56
//   l   = max(exp(psi(al))/be, fudge)       :58-65
//   term= -(a/b) e + lga + al (1 - log be) + lgamma(al)     :82-89
// other[k] is either given (other_nb == 0) or is the column sum of the OTHER side's block
// partials other_bp[other_nb][R+2], reduced here in the prologue.
// Block partials out: bp[block][0..R) = sum e per k, bp[block][R] = sum term, bp[block][R+1] = sum log l.
// ------------------------------------------------------------------------------------
// Device-resident driver loop (reference R/bayesian.R:336-352): the hyper-parameters, the lagging
// evidence lk0, the iteration counter and the stop flag live here; once `stop` is set every kernel
// of the remaining queued steps returns at once, so the state stays exactly as the break left it.
struct LoopCtl {
    double hyper[4];               // aw, bw, ah, bh (updated in place from step n0+1 on)
    double lk0, lkh;
    double stats[4];               // mean log lw, mean log lh, mean ew, mean eh of the last step
    double tol;
    int32_t it, stop, reason;      // reason: 1 NaN evidence, 2 converged, 3 hyper Newton failed, 4 Itmax
    int32_t max_it, n0, dn;
    int32_t flags[4];              // hyper.update
};

__device__ inline int dev_hyper_update_pair(const int32_t *flags, const double *stats, double *hyper, int lane);
__device__ __forceinline__ double block_sum(double s, double *sm);

// Device-driven loop of an unpartitioned engine: the control step (k_control below: the evidence of the step just swept,
// hyper_update, the stopping rule, the history row) FOLDED into the gene-side update of the NEXT step -- one launch
// and one launch gap less per step (k_control is 8.7 us of latency for a microsecond of work).  Every block of the
// update forms the control step for itself from the same inputs (the block partials of the two updates and the
// sweep's evidence partials: identical bits everywhere), so all blocks agree on the new hyper-parameters and on the stop;
// block 0 alone writes them out.  Nothing a block reads is written in the same launch: the control block and the
// gene-side block partials alternate between two buffers by step parity (prev / next, bpW_prev / the kernel's bp).
// After the last step of a run the same kernel is launched once more with control_only = 1 (one block).
struct ControlFold {
    const LoopCtl *prev;           // control block as the previous step left it (null: no fold)
    LoopCtl *next;                 // ... as this launch leaves it (written by block 0)
    const double *bpW_prev;        // [nb][R+2] block partials of the previous step's gene-side update
    const double *epart;           // the previous sweep's evidence partials
    int64_t nepart;
    double lgx, n, m_global;
    double *history, *out_host;
    int32_t do_control;            // 0: first step of a run (nothing to evaluate yet: next = prev)
    int32_t control_only;          // 1: the launch behind the last step (no update)
    // Cell-partitioned engines (the update is the dense one, on the all-reduced statistics): the cell side's sums come
    // all-reduced too -- tail_in = [rowSum(eh)_k | sum H-terms | sum log lh] (R + 2 doubles) --, epart / nepart are the
    // ELEMENT-WISE all-reduced evidence partials of every partition's sweeps (kEvSlots doubles, zero beyond the workgroups
    // in use) and lgx_in points at the all-reduced sum lgamma(x + 1); bpW_prev (nbW rows) is replicated.
    const double *tail_in;
    const double *lgx_in;
    int32_t nbW;
};

// Evidence partials of a partition's two sweeps as they travel through the all-reduce of a device-driven partitioned
// loop: a fixed number of slots (the same on every rank whatever its workgroup count), then sum lgamma(x + 1).
constexpr int kEvSlots = 1024;

#ifdef VBNMF_ABL_STAMPS                                       /* instrumented builds only (profiles/ubench/r04/update_stamps.sh) */
__device__ unsigned long long g_upd_stamps[2 * kUpdateBlocks * 12];
#define UPD_STAMP(i) do { if (threadIdx.x == 0 && !fold.control_only) g_upd_stamps[((size_t)side * kUpdateBlocks + blockIdx.x) * 12 + (i)] = wall_clock64(); } while (0)
#define UPD2_STAMP(i) do { if (threadIdx.x == 0) g_upd_stamps[((size_t)blockIdx.x) * 12 + (i)] = wall_clock64(); } while (0)   /* k_update2: the first table */
#else
#define UPD_STAMP(i) do { } while (0)
#define UPD2_STAMP(i) do { } while (0)
#endif

template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_update(
    const double *__restrict__ acc, const int32_t *__restrict__ inv_ptr, const uint32_t *__restrict__ inv_task,
    int64_t nmaj, int r, const double *__restrict__ other, const double *__restrict__ other_bp, int other_nb,
    double a, double b, double lga, double fudge,
    double *__restrict__ l, double *__restrict__ ll, double *__restrict__ e, double *__restrict__ d, double *__restrict__ bp,
    const LoopCtl *__restrict__ ctl, int side, const ControlFold fold, int stage_ids)
{
    constexpr int RB = kUpdateThreads / R;       // majors per pass
    __shared__ double s_other[R + 2];
    __shared__ double s_e[kUpdateThreads], s_t[kUpdateThreads], s_l[kUpdateThreads];
    __shared__ uint32_t s_ids[kStageIds];
    __shared__ int32_t s_ptr[kStagePtr];
    int stopped = 0;
    const int t = threadIdx.x;
    UPD_STAMP(0);
    // The block's stretch of the inverse index into LDS, first thing: two dependent global round trips (pointers, then
    // ids) that the prologue's own loads overlap with, instead of two per ROUND of every major's gather below.
    const int64_t per0 = (nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t bm0 = (int64_t)blockIdx.x * per0, bm1 = min(nmaj, bm0 + per0);
    int q_lo = 0;
    bool staged = false;
    if (stage_ids && inv_ptr && !fold.control_only && bm0 < bm1 && bm1 - bm0 < kStagePtr) {
        q_lo = inv_ptr[bm0];
        const int q_hi = inv_ptr[bm1];
        staged = q_hi - q_lo <= kStageIds;               // (block-uniform)
        if (staged) {
            for (int q = t; q <= (int)(bm1 - bm0); q += kUpdateThreads) s_ptr[q] = inv_ptr[bm0 + q];
            for (int q = q_lo + t; q < q_hi; q += kUpdateThreads) s_ids[q - q_lo] = inv_task[q];
        }
    }
    UPD_STAMP(1);
    if (fold.prev) {
        // ---- the folded control step (see ControlFold; the arithmetic is k_control's, statement by statement) ----
        __shared__ double sW[R + 2], s_hy[4];
        __shared__ int s_stop;
        const LoopCtl *pv = fold.prev;
        const int was_stopped = pv->stop;        // (tested below, once the loads of the reductions are in flight too)
        double part = 0.0;
        if (fold.do_control) for (int64_t q = t; q < fold.nepart; q += kUpdateThreads) part += fold.epart[q];
        // column sums of the previous gene-side partials and of the cell-side ones: the latter are also rowSums(eh),
        // the `other` of this update
        if (fold.tail_in) {
            bp_colsums(fold.bpW_prev, fold.nbW, R + 2, sW, kUpdateThreads);
            if (t < R + 2) s_other[t] = fold.tail_in[t];
        } else {
            bp_colsums2(fold.bpW_prev, other_bp, other_nb, R + 2, sW, s_other, kUpdateThreads);
        }
        UPD_STAMP(8);
        if (was_stopped) {                       // a step queued past the stop: the control block and this block's row of the
            if (blockIdx.x == 0 && t == 0) *fold.next = *pv;                  // gene-side partials travel on unchanged
            if (t < R + 2 && !fold.control_only) bp[(size_t)blockIdx.x * (R + 2) + t] = fold.bpW_prev[(size_t)blockIdx.x * (R + 2) + t];
            return;
        }
        const double data = block_sum(part, s_e);            // (two barriers: sW, s_other are complete behind it)
        UPD_STAMP(9);
        if (t < 2) {                              // lanes 0 and 1: the two Newton recurrences of hyper_update side by side
            if (t == 0) for (int q = 0; q < 4; q++) s_hy[q] = pv->hyper[q];
            int reason = 0, it = pv->it;
            double lkh = pv->lkh, new_lk0 = pv->lk0;
            double st[4] = {pv->stats[0], pv->stats[1], pv->stats[2], pv->stats[3]};
            if (fold.do_control) {
                double cross = 0.0, sew = 0.0, seh = 0.0;
                for (int k = 0; k < r; k++) { cross += sW[k] * s_other[k]; sew += sW[k]; seh += s_other[k]; }
                const double lgx = fold.lgx_in ? *fold.lgx_in : fold.lgx;
                const double U = -cross - data - lgx + sW[R] + s_other[R];
                lkh = U / (fold.n * fold.m_global);
                st[0] = sW[R + 1] / (fold.n * r); st[1] = s_other[R + 1] / (fold.m_global * r);
                st[2] = sew / (fold.n * r); st[3] = seh / (fold.m_global * r);
                it = pv->it + 1;
                if (it > pv->n0 && it % pv->dn == 0) {
                    if (dev_hyper_update_pair(pv->flags, st, s_hy, t)) reason = 3;
                }
                if (t == 0 && !reason) {
                    const double lk0 = pv->lk0;
                    if (lkh != lkh) reason = 1;
                    else if (it > 1 && it > pv->n0 && lkh >= lk0 && fabs(1.0 - lkh / lk0) < pv->tol) reason = 2;
                    else { new_lk0 = lkh; if (it >= pv->max_it) reason = 4; }
                }
            }
            if (t == 0) {
                s_stop = reason != 0;
                if (blockIdx.x == 0) {
                    LoopCtl nx = *pv;
                    nx.it = it; nx.lkh = lkh; nx.lk0 = new_lk0;
                    for (int q = 0; q < 4; q++) { nx.stats[q] = st[q]; nx.hyper[q] = s_hy[q]; }
                    if (reason) { nx.reason = reason; nx.stop = 1; }
                    *fold.next = nx;
                    if (fold.do_control) {
                        if (fold.history) {
                            double *h = fold.history + (size_t)(it - 1) * 9;
                            h[0] = lkh;
                            for (int q = 0; q < 4; q++) { h[1 + q] = st[q]; h[5 + q] = s_hy[q]; }
                        }
                        double *oh = fold.out_host;
                        oh[0] = lkh;
                        for (int q = 0; q < 4; q++) { oh[1 + q] = st[q]; oh[8 + q] = s_hy[q]; }
                        oh[12] = new_lk0;
                        oh[5] = (double)it;
                        __threadfence_system();
                        reinterpret_cast<volatile double *>(oh)[6] = (double)reason;
                        reinterpret_cast<volatile double *>(oh)[7] = (double)it;
                    }
                }
            }
        }
        UPD_STAMP(10);
        __syncthreads();
        if (fold.control_only) return;
        if (s_stop) {                            // the loop ends here: no update, the partials' row travels on (as above)
            if (t < R + 2) bp[(size_t)blockIdx.x * (R + 2) + t] = fold.bpW_prev[(size_t)blockIdx.x * (R + 2) + t];
            return;
        }
        a = s_hy[2 * side]; b = s_hy[2 * side + 1];
    } else {
    if (ctl) {                                   // device-driven loop: hyper-parameters come from the control block
        stopped = ctl->stop;                     // (tested below: these loads and the column sums' travel together)
        a = ctl->hyper[2 * side]; b = ctl->hyper[2 * side + 1];
    }
    if (other_nb > 0) bp_colsums(other_bp, other_nb, R + 2, s_other, kUpdateThreads);
    else if (t < R) s_other[t] = other[t];
    if (stopped) return;
    __syncthreads();
    }
    UPD_STAMP(2);
    if (ctl || fold.prev) {
        double psi_a, lg_a;
        dev_psi_lgamma(a, &psi_a, &lg_a);
        lga = -lg_a + a * log(a / b);            // reference src/vbnmf_update.cpp:82 / :87
    }

    const int row = t / R, k = t - row * R;
    const int64_t per = (nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = min(nmaj, m0 + per);
    const double be = a / b + s_other[k < R ? k : 0];
    const double lbe = log(be);
    double ve = 0.0, vt = 0.0, vl = 0.0;
    UPD_STAMP(3);
    if (row < RB) {
        for (int64_t M = m0 + row; M < m1; M += RB) {
            const size_t o = (size_t)M * R + k;
            if (k < r) {
#ifdef VBNMF_ABL_NOGATHER                                    /* ablation builds only (profiles/ubench/r04/upd_ablate.sh) */
                const double s = 1.0 + 1e-9 * (double)k;
#else
                const double s = !inv_ptr ? acc[o]
                                 : staged ? task_sum_lds(acc, s_ids, s_ptr[M - bm0] - q_lo, s_ptr[M - bm0 + 1] - q_lo, R, k)
                                          : task_sum(acc, inv_task, inv_ptr[M], inv_ptr[M + 1], R, k);
#endif
                const double al = a + l[o] * s;
                const double ev = al / be;
                const double dv = al / be / be;
                double psi, lgam;
#ifdef VBNMF_ABL_NOSPECIAL
                psi = al * 0.5; lgam = al * 0.25;
                const double tmp = psi / be;
                const double ln = (tmp > fudge ? tmp : fudge);
                const double lg = ln * 0.125;
#else
                dev_psi_lgamma(al, &psi, &lgam);
                if (!(al > 0.0)) { psi = __builtin_nan(""); lgam = __builtin_nan(""); }
                const double tmp = exp(psi) / be;
                const double ln = (tmp > fudge ? tmp : fudge);
                const double lg = log(ln);
#endif
                ve += ev;
                vt += -(a / b) * ev + lga + al * (1.0 - lbe) + lgam;
                vl += lg;
#ifdef VBNMF_ABL_NOWRITE
                if (ln == 123.456) { l[o] = ln; ll[o] = ln * lg; e[o] = ev; d[o] = dv; }
#else
                l[o] = ln; ll[o] = ln * lg; st_once(e + o, ev); st_once(d + o, dv);
#endif
            } else {
                l[o] = 0.0; ll[o] = 0.0; e[o] = 0.0; d[o] = 0.0;
            }
        }
    }
    UPD_STAMP(4);
    s_e[t] = ve; s_t[t] = vt; s_l[t] = vl;
    __syncthreads();
    UPD_STAMP(5);
    constexpr int P2 = (RB <= 32) ? 32 : (RB <= 64) ? 64 : (RB <= 128) ? 128 : (RB <= 256) ? 256 : 512;
    for (int h = P2 / 2; h >= 1; h >>= 1) {
        if (row < h && row + h < RB) { s_e[t] += s_e[t + h * R]; s_t[t] += s_t[t + h * R]; s_l[t] += s_l[t + h * R]; }
        __syncthreads();
    }
    UPD_STAMP(6);
    double *o = bp + (size_t)blockIdx.x * (R + 2);
    if (t < R) o[t] = s_e[t];
    if (t == 0) {
        double st = 0.0, sl = 0.0;
        for (int q = 0; q < R; q++) { st += s_t[q]; sl += s_l[q]; }
        o[R] = st; o[R + 1] = sl;
    }
    UPD_STAMP(7);
}

// ------------------------------------------------------------------------------------
// Both posterior updates in ONE launch (unpartitioned engines; round 5).  The H update's rate needs colSums(ew_new)
// (reference src/vbnmf_update.cpp:53 -- `ew` is updated before `beh`, :44 before :53), which used to force a kernel
// boundary between the two updates.  But sum_i ew_ik = sum_i (aw + sw_ik) / bew_k = (n aw + sum_i sw_ik) / bew_k, and
// the gene side of the sweep now leaves sum_i sw_ik (SweepSide::csum: per-slice sums by a fixed tree, added per
// workgroup in list order -- no atomics, bit-reproducible run to run), so every block knows both rates from the start:
//   bew_k = aw/bw + rowSums(eh_in)_k                    :40-43   (block partials of the previous H update)
//   beh_k = ah/bh + (n aw + sum_i sw_ik) / bew_k        :50-53   (the NEW ew's column sums, as the reference orders it)
// and works through its genes and its cells with no hand-off in between.  The gene side's results are the two-launch
// form's bit for bit; beh_k differs from it by the rounding of one sum (the two-launch form adds the n rounded quotients
// ew_ik; this one divides the sum): ~1e-16 relative.  The evidence's cross term keeps using the sum of the stored ew (bpW).
//
// The block's work is a TABLE the host cuts when the engine is made (engine.hip: build_update_table), one row of
// `stride4` 16-byte words per block, the same shape for every block, so its load heads the kernel and depends on nothing:
//   visits  [RB][V][3]   thread row `row` makes visits (row, 0), (row, 1), ...: {side << 31 | major - the block's first major
//                        of that side, q0, q1} = the major's stretch of the ids below; 0xFFFFFFFF ends a row's list.
//                        The block's majors of BOTH sides are dealt to the rows longest-processing-time first (cost = rounds of
//                        16 task rows + the posterior's arithmetic): a gene with 80 tasks no longer holds its block
//                        while the other rows idle (k_update walks the majors in index order: in-kernel stamps, rank 10:
//                        thread 0 done at 23.6 us, the block's last thread at 33).
//   ids     [..]         the task ids of the block's majors (the inverse index's stretches, copied)
// The time line of the first version (index order, the inverse index staged through four dependent round trips; rank 10,
// profiles/r05_update2_stamps.txt): 6.0 us until the staging loads are out, 3.8 column sums, 0.9 + 3.6 control step,
// 11.8 + 11.8 the two stretches of thread 0, 9.3 waiting for the block's slowest thread, 1.7 reduction = 49 us.
// Block partials out: W.bp[block][0..R+2), H.bp[block][0..R+2) as k_update; both tables alternate by step (nothing a
// block reads is written in the same launch).
// ------------------------------------------------------------------------------------
struct UpdSide {
    const double *part;            // [n_tasks][R] per-task partial statistics of this side's sweep
    int64_t nmaj;
    double *l, *ll, *e, *d;        // state arrays of the factor
    double *bp;                    // [nb][R+2] block partials written by this launch
    const double *bp_prev;         // ... of the previous update of this factor (H: rowSums(eh_in); W: the control step's colSums(ew))
};
struct UpdTable {
    const uint4 *tab;              // [blocks][stride4]
    int32_t stride4;               // 16-byte words per block (<= kUpdTabWords / 4)
    int32_t V;                     // visits per thread row (the longest list)
    int32_t ids_off;               // word offset of the ids inside a block's row
};
constexpr int kUpdTabWords = 16384;        // LDS copy of a block's row of the table: 64 KB


// The visits of one thread row to the majors of one side (the row's list holds its genes first, then its cells): column k
// of every visited major -- gather of the task partials, Gamma posterior, geometric mean, evidence terms (k_update's body).
template <int R>
__device__ __forceinline__ int posterior_visits(const UpdSide &S, int64_t m0, const uint32_t *vis, int v, int V, uint32_t side,
                                                const uint32_t *ids, int k, int r, double a, double ab, double be, double lbe,
                                                double lga, double fudge, double &ve, double &vt, double &vl)
{
    for (; v < V; v++) {
        const uint32_t code = vis[3 * v];
        if (code == 0xFFFFFFFFu || (code >> 31) != side) break;
        const int64_t M = m0 + (int64_t)(code & 0x7FFFFFFFu);
        const int q0 = (int)vis[3 * v + 1], q1 = (int)vis[3 * v + 2];
        const size_t o = (size_t)M * R + k;
        if (k < r) {
            const double s = task_sum_lds(S.part, ids, q0, q1, R, k);
            const double al = a + S.l[o] * s;
            const double ev = al / be;
            const double dv = al / be / be;
            double psi, lgam;
            dev_psi_lgamma(al, &psi, &lgam);
            if (!(al > 0.0)) { psi = __builtin_nan(""); lgam = __builtin_nan(""); }
            const double tmp = exp(psi) / be;
            const double ln = (tmp > fudge ? tmp : fudge);
            const double lg = log(ln);
            ve += ev;
            vt += -ab * ev + lga + al * (1.0 - lbe) + lgam;
            vl += lg;
            S.l[o] = ln; S.ll[o] = ln * lg; st_once(S.e + o, ev); st_once(S.d + o, dv);
        } else {
            S.l[o] = 0.0; S.ll[o] = 0.0; S.e[o] = 0.0; S.d[o] = 0.0;
        }
    }
    return v;
}

template <int R>
__device__ __forceinline__ void update2_body(
    const UpdSide &W, const UpdSide &H, const UpdTable &T, int r, int nb, const double *__restrict__ csum, int ncs,
    double aw, double bw, double ah, double bh, double fudge, const LoopCtl *__restrict__ ctl, const ControlFold &fold)
{
    constexpr int RB = kUpdateThreads / R;
    __shared__ double s_other[R + 2], s_cs[R + 2], sW[R + 2], s_hy[4];
    __shared__ double s_red[6][kUpdateThreads];
    __shared__ uint4 s_tab4[kUpdTabWords / 4];
    __shared__ int s_stop;
    const uint32_t *s_tab = reinterpret_cast<const uint32_t *>(s_tab4);
    const int t = threadIdx.x;
    UPD2_STAMP(0);
    // the block's row of the table: loads issued first thing, parked in registers, written to LDS behind the other loads
    uint4 tb[4];
    {
        const uint4 *src = T.tab + (size_t)blockIdx.x * T.stride4;
#pragma unroll
        for (int u = 0; u < 4; u++) { const int q = t + u * kUpdateThreads; if (q < T.stride4) tb[u] = src[q]; }
    }
    const int64_t perW = (W.nmaj + gridDim.x - 1) / gridDim.x, perH = (H.nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t w0 = min(W.nmaj, (int64_t)blockIdx.x * perW), h0 = min(H.nmaj, (int64_t)blockIdx.x * perH);
    int stopped = 0;
    UPD2_STAMP(1);
    if (fold.prev) {
        // ---- the folded control step (k_update's, statement by statement; the cell side's sums come from H.bp_prev) ----
        const LoopCtl *pv = fold.prev;
        const int was_stopped = pv->stop;
        double part = 0.0;
        if (fold.do_control) for (int64_t q = t; q < fold.nepart; q += kUpdateThreads) part += fold.epart[q];
        bp_colsums3(W.bp_prev, H.bp_prev, nb, R + 2, csum, ncs, R, sW, s_other, s_cs, kUpdateThreads);
        UPD2_STAMP(8);
        if (was_stopped) {                       // a step queued past the stop: the control block and both partial tables travel on
            if (blockIdx.x == 0 && t == 0) *fold.next = *pv;
            if (t < R + 2) {
                W.bp[(size_t)blockIdx.x * (R + 2) + t] = W.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
                H.bp[(size_t)blockIdx.x * (R + 2) + t] = H.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int q = t + u * kUpdateThreads; if (q < T.stride4) s_tab4[q] = tb[u]; }
        const double data = block_sum(part, s_red[0]);       // (two barriers: sW, s_other, s_cs and the table are complete behind it)
        UPD2_STAMP(9);
        if (t < 2) {
            if (t == 0) for (int q = 0; q < 4; q++) s_hy[q] = pv->hyper[q];
            int reason = 0, it = pv->it;
            double lkh = pv->lkh, new_lk0 = pv->lk0;
            double st[4] = {pv->stats[0], pv->stats[1], pv->stats[2], pv->stats[3]};
            if (fold.do_control) {
                double cross = 0.0, sew = 0.0, seh = 0.0;
                for (int k = 0; k < r; k++) { cross += sW[k] * s_other[k]; sew += sW[k]; seh += s_other[k]; }
                const double U = -cross - data - fold.lgx + sW[R] + s_other[R];
                lkh = U / (fold.n * fold.m_global);
                st[0] = sW[R + 1] / (fold.n * r); st[1] = s_other[R + 1] / (fold.m_global * r);
                st[2] = sew / (fold.n * r); st[3] = seh / (fold.m_global * r);
                it = pv->it + 1;
                if (it > pv->n0 && it % pv->dn == 0) {
                    if (dev_hyper_update_pair(pv->flags, st, s_hy, t)) reason = 3;
                }
                if (t == 0 && !reason) {
                    const double lk0 = pv->lk0;
                    if (lkh != lkh) reason = 1;
                    else if (it > 1 && it > pv->n0 && lkh >= lk0 && fabs(1.0 - lkh / lk0) < pv->tol) reason = 2;
                    else { new_lk0 = lkh; if (it >= pv->max_it) reason = 4; }
                }
            }
            if (t == 0) {
                s_stop = reason != 0;
                if (blockIdx.x == 0) {
                    LoopCtl nx = *pv;
                    nx.it = it; nx.lkh = lkh; nx.lk0 = new_lk0;
                    for (int q = 0; q < 4; q++) { nx.stats[q] = st[q]; nx.hyper[q] = s_hy[q]; }
                    if (reason) { nx.reason = reason; nx.stop = 1; }
                    *fold.next = nx;
                    if (fold.do_control) {
                        if (fold.history) {
                            double *h = fold.history + (size_t)(it - 1) * 9;
                            h[0] = lkh;
                            for (int q = 0; q < 4; q++) { h[1 + q] = st[q]; h[5 + q] = s_hy[q]; }
                        }
                        double *oh = fold.out_host;
                        oh[0] = lkh;
                        for (int q = 0; q < 4; q++) { oh[1 + q] = st[q]; oh[8 + q] = s_hy[q]; }
                        oh[12] = new_lk0;
                        oh[5] = (double)it;
                        __threadfence_system();
                        reinterpret_cast<volatile double *>(oh)[6] = (double)reason;
                        reinterpret_cast<volatile double *>(oh)[7] = (double)it;
                    }
                }
            }
        }
        UPD2_STAMP(10);
        __syncthreads();
        if (s_stop) {                            // the loop ends here: no update, the partials' rows travel on (as above)
            if (t < R + 2) {
                W.bp[(size_t)blockIdx.x * (R + 2) + t] = W.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
                H.bp[(size_t)blockIdx.x * (R + 2) + t] = H.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
            }
            return;
        }
        aw = s_hy[0]; bw = s_hy[1]; ah = s_hy[2]; bh = s_hy[3];
    } else {
        if (ctl) {                               // device-driven loop without the fold: hyper-parameters from the control block
            stopped = ctl->stop;
            aw = ctl->hyper[0]; bw = ctl->hyper[1]; ah = ctl->hyper[2]; bh = ctl->hyper[3];
        }
        bp_colsums3(nullptr, H.bp_prev, nb, R + 2, csum, ncs, R, nullptr, s_other, s_cs, kUpdateThreads);
        if (stopped) {                           // (both tables alternate by launch: the rows travel on)
            if (t < R + 2) {
                W.bp[(size_t)blockIdx.x * (R + 2) + t] = W.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
                H.bp[(size_t)blockIdx.x * (R + 2) + t] = H.bp_prev[(size_t)blockIdx.x * (R + 2) + t];
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int q = t + u * kUpdateThreads; if (q < T.stride4) s_tab4[q] = tb[u]; }
        __syncthreads();
    }
    UPD2_STAMP(2);
    double psi_a, lg_a;
    dev_psi_lgamma(aw, &psi_a, &lg_a);
    const double lgaW = -lg_a + aw * log(aw / bw);   // reference src/vbnmf_update.cpp:82
    dev_psi_lgamma(ah, &psi_a, &lg_a);
    const double lgaH = -lg_a + ah * log(ah / bh);   // :87

    const int row = t / R, k = t - row * R;
    const int kc = k < R ? k : 0;
    const double beW = aw / bw + s_other[kc];                                   // :40-43
    const double beH = ah / bh + ((double)W.nmaj * aw + s_cs[kc]) / beW;        // :50-53 on the NEW ew's column sums
    const double lbeW = log(beW), lbeH = log(beH);
    double acc6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // sum e, sum term, sum log l of the gene side; the same of the cell side
    UPD2_STAMP(3);
    if (row < RB) {
        // a row's visits to genes come first, its visits to cells behind them: inside each loop the factor's arrays are the
        // same for every lane (scalar base addresses)
        const uint32_t *vis = s_tab + (size_t)row * T.V * 3;
        const uint32_t *ids = s_tab + T.ids_off;
        int v = 0;
        v = posterior_visits<R>(W, w0, vis, v, T.V, 0u, ids, k, r, aw, aw / bw, beW, lbeW, lgaW, fudge, acc6[0], acc6[1], acc6[2]);
        UPD2_STAMP(11);
        v = posterior_visits<R>(H, h0, vis, v, T.V, 1u, ids, k, r, ah, ah / bh, beH, lbeH, lgaH, fudge, acc6[3], acc6[4], acc6[5]);
    }
    UPD2_STAMP(4);
#pragma unroll
    for (int q = 0; q < 6; q++) s_red[q][t] = acc6[q];
    __syncthreads();
    UPD2_STAMP(5);
    constexpr int P2 = (RB <= 32) ? 32 : (RB <= 64) ? 64 : (RB <= 128) ? 128 : (RB <= 256) ? 256 : 512;
    for (int h = P2 / 2; h >= 1; h >>= 1) {
        if (row < h && row + h < RB) {
#pragma unroll
            for (int q = 0; q < 6; q++) s_red[q][t] += s_red[q][t + h * R];
        }
        __syncthreads();
    }
    UPD2_STAMP(6);
    double *oW = W.bp + (size_t)blockIdx.x * (R + 2), *oH = H.bp + (size_t)blockIdx.x * (R + 2);
    if (t < R) { oW[t] = s_red[0][t]; oH[t] = s_red[3][t]; }
    if (t == 0) {
        double st = 0.0, sl = 0.0;
        for (int q = 0; q < R; q++) { st += s_red[1][q]; sl += s_red[2][q]; }
        oW[R] = st; oW[R + 1] = sl;
    }
    if (t == 64) {
        double st = 0.0, sl = 0.0;
        for (int q = 0; q < R; q++) { st += s_red[4][q]; sl += s_red[5][q]; }
        oH[R] = st; oH[R + 1] = sl;
    }
    UPD2_STAMP(7);
}

template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_update2(
    const UpdSide W, const UpdSide H, const UpdTable T, int r, int nb, const double *__restrict__ csum, int ncs,
    double aw, double bw, double ah, double bh, double fudge, const LoopCtl *__restrict__ ctl, const ControlFold fold)
{
    update2_body<R>(W, H, T, r, nb, csum, ncs, aw, bw, ah, bh, fudge, ctl, fold);
}

// ------------------------------------------------------------------------------------
// A BATCH of engines stepped by one launch (vbnmf_batch_run): the restarts of one rank on one matrix -- same layouts, same
// table, same grids, each with its own state, partials and control block -- take one row of the grid each (blockIdx.y).  On the
// matrices the reference ships (1030 x 450) a step is two dependent launches of ~15 us for microseconds of work on a dozen
// workgroups, and the device runs the launches of different streams (let alone processes) mostly one after the other
// (profiles/r05_small_concurrent.txt): the independent units of a rank sweep share a launch instead.  The bodies are the
// single engine's, so every unit's results are its stand-alone results bit for bit.
// ------------------------------------------------------------------------------------
struct Upd2Job {
    UpdSide W, H;
    UpdTable T;
    int32_t r, nb, ncs;
    const double *csum;
    double fudge;
    ControlFold fold;              // (the batch always runs the device-driven loop with the control step folded in)
};

template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_update2_batch(const Upd2Job *__restrict__ jobs)
{
    const Upd2Job J = jobs[blockIdx.y];
    update2_body<R>(J.W, J.H, J.T, J.r, J.nb, J.csum, J.ncs, 0.0, 0.0, 0.0, 0.0, J.fudge, nullptr, J.fold);
}

// State load (set_state): ll = l*log(l) and the block partials of e's column sums, same
// bp layout as k_update (term and log-sum columns are written as 0).
template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_prime(int64_t nmaj, int r, const double *__restrict__ l,
                                                          double *__restrict__ ll, const double *__restrict__ e,
                                                          double *__restrict__ bp)
{
    constexpr int RB = kUpdateThreads / R;
    __shared__ double s_e[kUpdateThreads];
    const int t = threadIdx.x;
    const int row = t / R, k = t - row * R;
    const int64_t per = (nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = min(nmaj, m0 + per);
    double ve = 0.0;
    if (row < RB) {
        for (int64_t M = m0 + row; M < m1; M += RB) {
            const size_t o = (size_t)M * R + k;
            if (k < r) { const double v = l[o]; ll[o] = v * log(v); if (e) ve += e[o]; }
            else ll[o] = 0.0;
        }
    }
    s_e[t] = ve;
    __syncthreads();
    constexpr int P2 = (RB <= 32) ? 32 : (RB <= 64) ? 64 : (RB <= 128) ? 128 : (RB <= 256) ? 256 : 512;
    for (int h = P2 / 2; h >= 1; h >>= 1) {
        if (row < h && row + h < RB) s_e[t] += s_e[t + h * R];
        __syncthreads();
    }
    double *o = bp + (size_t)blockIdx.x * (R + 2);
    if (t < R) o[t] = s_e[t];
    if (t == 0) { o[R] = 0.0; o[R + 1] = 0.0; }
}

// ------------------------------------------------------------------------------------
// Cell-partitioned runs: the gene-side statistics and the cell-side scalars go into the
// reduce buffer [swsum n*R | tail R+4] that the caller all-reduces.
// ------------------------------------------------------------------------------------
// out[major][k] = sum of the major's task partials, fixed order.
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ part, const int32_t *__restrict__ inv_ptr,
                                              const uint32_t *__restrict__ inv_task, int64_t nmaj, int R,
                                              double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nmaj * R) return;
    const int64_t M = e / R;
    const int k = (int)(e - M * R);
    out[e] = task_sum(part, inv_task, inv_ptr[M], inv_ptr[M + 1], R, k);
}

// k_pack and k_tail_h in one launch (the device-driven partitioned loop): one extra block forms the cell side's column sums.
__global__ __launch_bounds__(256) void k_pack_tail(const double *__restrict__ part, const int32_t *__restrict__ inv_ptr,
                                                   const uint32_t *__restrict__ inv_task, int64_t nmaj, int R,
                                                   double *__restrict__ out, const double *__restrict__ bpH, int nbH,
                                                   const int32_t *__restrict__ stop)
{
    if (blockIdx.x == gridDim.x - 1) {
        if (stop && *stop) return;                         // (as k_tail_h: steps queued past the stop leave the tail as it is)
        bp_colsums(bpH, nbH, R + 2, out + nmaj * R, 256);
        return;
    }
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nmaj * R) return;
    const int64_t M = e / R;
    const int k = (int)(e - M * R);
    out[e] = task_sum(part, inv_task, inv_ptr[M], inv_ptr[M + 1], R, k);
}

// Device-driven loop of a cell-partitioned engine: the reduce buffer is sent in two pieces.  The first,
// [swsum n*R | rowSum(eh)_k (R) | sum H-terms | sum log lh], is complete once the gene-side sweep and the H update are
// done (k_pack + k_tail_h) and travels while the cell-side sweep runs; the second, [data term | sum lgamma(x+1)], needs
// both sweeps (k_tail_data) and is two doubles.
__global__ __launch_bounds__(1024) void k_tail_h(const double *__restrict__ bpH, int nbH, int R, double *__restrict__ tail,
                                                 const int32_t *__restrict__ stop)
{
    if (stop && *stop) return;
    bp_colsums(bpH, nbH, R + 2, tail, 1024);
}

// In-process stand-in for the all-reduce between partition engines that share one device (tests, single-GPU
// rehearsals of a partitioned run): recv[p][i] = send[0][i] + send[1][i] + ... in partition order, for every p.
__global__ __launch_bounds__(256) void k_group_sum(const double *const *__restrict__ send, double *const *__restrict__ recv,
                                                   int parts, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = send[0][i];
    for (int p = 1; p < parts; p++) s += send[p][i];
    for (int p = 0; p < parts; p++) recv[p][i] = s;
}

// Sum of v[0..count) by one 1024-thread block, fixed order: thread t adds t, t+1024, ...; a shuffle tree per wave; the
// 16 wave sums through LDS and one more shuffle tree.  Two barriers (the ten-round LDS tree it replaces cost k_control
// and k_final about a microsecond each).
__device__ __forceinline__ double block_sum(double s, double *sm);
__device__ __forceinline__ double block_vec_sum(const double *__restrict__ v, int64_t count, double *sm)
{
    double s = 0.0;
    for (int64_t q = threadIdx.x; q < count; q += 1024) s += v[q];
    return block_sum(s, sm);
}
// the block's total of one value per thread (1024 threads), same order as above
__device__ __forceinline__ double block_sum(double s, double *sm)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    s = wave_sum(s);
    if (lane == 0) sm[wave] = s;
    __syncthreads();
    double r = lane < 16 ? sm[lane] : 0.0;
    r = wave_sum(r);                                  // every wave forms the same total in its lane 0 ...
    r = __shfl(r, 0, 64);                             // ... and hands it to all its lanes
    __syncthreads();                                  // sm may be reused by the caller
    return r;
}

// tail = [rowSum(eh)_k (R) | sum H-terms | sum log lh | data term | sum lgamma(x+1)] of THIS partition.
__global__ __launch_bounds__(1024) void k_tail(const double *__restrict__ bpH, int nbH, int R,
                                               const double *__restrict__ epart, int64_t nepart, double lgx,
                                               double *__restrict__ tail)
{
    __shared__ double sm[1024];
    bp_colsums(bpH, nbH, R + 2, tail, 1024);
    const double data = block_vec_sum(epart, nepart, sm);
    if (threadIdx.x == 0) { tail[R + 2] = data; tail[R + 3] = lgx; }
}

__global__ __launch_bounds__(1024) void k_tail_data(const double *__restrict__ epart, int64_t nepart, double lgx,
                                                    double *__restrict__ out2, const int32_t *__restrict__ stop)
{
    __shared__ double sm[1024];
    if (stop && *stop) return;
    const double data = block_vec_sum(epart, nepart, sm);
    if (threadIdx.x == 0) { out2[0] = data; out2[1] = lgx; }
}

// Evidence and the four hyper statistics.  One block.
//   W side : colSum(ew)_k, sum W-terms, sum log lw  = column sums of bpW (replicated in every partition)
//   tail   : as k_tail writes it; either given (summed over partitions by the caller) or, when
//            tail == nullptr, formed here from bpH / epart / lgx (single-GPU path, no extra launch).
//   out    : lkh, mean log lw, mean log lh, mean ew, mean eh, then out[7] = seq (the host polls it).
template <int R>
__global__ __launch_bounds__(1024) void k_final(const double *__restrict__ bpW, int nbW, const double *__restrict__ tail_in,
                                                const double *__restrict__ bpH, int nbH,
                                                const double *__restrict__ epart, int64_t nepart, double lgx,
                                                int r, double n, double m_global, double seq, double *__restrict__ out,
                                                double *__restrict__ out_host)
{
    __shared__ double sW[R + 2], sT[R + 4];
    __shared__ double sm[1024];
    if (tail_in) {
        bp_colsums(bpW, nbW, R + 2, sW, 1024);
        if (threadIdx.x < R + 4) sT[threadIdx.x] = tail_in[threadIdx.x];
        __syncthreads();
    } else {
        double part = 0.0;                       // the three reductions' loads travel together
        for (int64_t q = threadIdx.x; q < nepart; q += 1024) part += epart[q];
        bp_colsums2(bpW, bpH, nbW, R + 2, sW, sT, 1024);
        const double data = block_sum(part, sm);
        if (threadIdx.x == 0) { sT[R + 2] = data; sT[R + 3] = lgx; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double cross = 0.0, sew = 0.0, seh = 0.0;
        for (int k = 0; k < r; k++) { cross += sW[k] * sT[k]; sew += sW[k]; seh += sT[k]; }
        const double U = -cross - sT[R + 2] - sT[R + 3] + sW[R] + sT[R];
        double o[5];
        o[0] = U / (n * m_global);
        o[1] = sW[R + 1] / (n * r);
        o[2] = sT[R + 1] / (m_global * r);
        o[3] = sew / (n * r);
        o[4] = seh / (m_global * r);
        for (int q = 0; q < 5; q++) { out[q] = o[q]; out_host[q] = o[q]; }
        __threadfence_system();
        reinterpret_cast<volatile double *>(out_host)[7] = seq;
    }
}

// ------------------------------------------------------------------------------------
// Device-resident driver loop: hyper_update (reference R/bayesian.R:2-53) and the loop control of
// vb_iterate (R/bayesian.R:342-348) after each step, by one thread.
// ------------------------------------------------------------------------------------
__device__ inline double dev_trigamma(double x)
{
    double s = 0.0;
    while (x < 10.0) { s += 1.0 / (x * x); x += 1.0; }
    const double xi = 1.0 / x, y = xi * xi;
    const double ser = xi * y * (1.0 / 6 - y * (1.0 / 30 - y * (1.0 / 42 - y * (1.0 / 30 - y * (5.0 / 66 - y * (691.0 / 2730 - y * (7.0 / 6)))))));
    return s + xi + 0.5 * y + ser;
}
__device__ inline double dev_digamma1(double x) { double p, g; dev_psi_lgamma(x, &p, &g); return (x > 0.0) ? p : __builtin_nan(""); }

// hyper_update (R/bayesian.R:2-53) by lanes 0 and 1 of a wave side by side: lane 0 iterates aw, lane 1 ah -- the two
// Newton recurrences (:19-27) only meet in the convergence test (:37), which the lanes form by exchanging their
// terms, so both run the same number of iterations as the sequential form and reach the same values.  A dependent fp64
// chain (log, digamma, trigamma) costs microseconds on one lane; this halves it.
// Returns 0, or 1 when the Newton iteration does not converge ("Hyper-parameter update failed to converge", :43);
// lane 0 stores the four new hyper-parameters.
__device__ inline int dev_hyper_update_pair(const int32_t *flags, const double *stats, double *hyper, int lane)
{
    if (flags[0] + flags[1] + flags[2] + flags[3] == 0) return 0;                          // :4
    double a0 = hyper[2 * lane];                           // aw (lane 0) or ah (lane 1)
    const double b0 = hyper[2 * lane + 1];                 // bw or bh
    const double lm = lane ? stats[1] : stats[0], em = lane ? stats[3] : stats[2];   // mean log l, mean e of this side (:8-11)
    const int fa = flags[2 * lane];
    double a1 = a0;
    int failed = 0;
    if (flags[0] + flags[2] > 0) {                                                         // :15
        int i = 1;
        while (i < 100) {                                                                  // Niter = 100 (:343)
            double d = fa ? (log(a0) - dev_digamma1(a0) - em / b0 + 1.0 + lm - log(b0)) / (1.0 / a0 - dev_trigamma(a0)) : 0.0;
            // A non-finite step (a statistic of -inf: fudge = 0 with a shape so small that exp(psi) underflows to 0)
            // would make the reference's halving loop below spin for ever -- on the host there, on the GPU here.
            // It is reported as a failed hyper-parameter update (reason 3) instead; the halvings are bounded too
            // (a finite d reaches a1 > 0 in < 1100 of them).
            int bad = !(fabs(d) <= 1.79769313486231570815e308);
            a1 = a0 - d;
            for (int hv = 0; !bad && a1 <= 0.0; hv++) {                                    // :28-35
                d *= 0.5; a1 = a0 - d;
                if (hv >= 1100) bad = 1;
            }
            const double u = bad ? 0.0 : 1.0 - a1 / a0;
            const double uw = __shfl(u, 0, 64), uh = __shfl(u, 1, 64);
            bad = __shfl(bad, 0, 64) | __shfl(bad, 1, 64);
            if (bad) { failed = 1; break; }
            if (uw * uw + uh * uh < 1e-3) break;                                           // Tol = 1e-3 (:344)
            a0 = a1; i++;
        }
        if (i == 100) failed = 1;
    }
    const double aw1 = __shfl(a1, 0, 64), ah1 = __shfl(a1, 1, 64);
    if (lane == 0 && !failed) {
        hyper[0] = aw1; hyper[1] = flags[1] ? stats[2] : hyper[1];                         // :48-49
        hyper[2] = ah1; hyper[3] = stats[3];                                               // :50-51 (both branches assign ehm)
    }
    return failed;
}

// One block after each sweep of a device-driven loop: the reductions of k_final, then
//   it <- it+1 ; hyper_update if it > n0 and it %% dn == 0 (:342-344) ; break on NaN (:345) ;
//   break if it > 1, it > n0, lkh >= lk0 and |1 - lkh/lk0| < Tol (:346-347, lk0 NOT refreshed) ; lk0 <- lkh (:348).
// history row it-1 = [lkh, 4 statistics, 4 hyper-parameters after the update]; out_host = [lkh, 4 stats, it, reason, it].
// Cell-partitioned engines hand in the all-reduced pieces instead: tail_in = [rowSum(eh)_k | sum H-terms | sum log lh]
// (R + 2 doubles) and small_in = [data term | sum lgamma(x+1)], both summed over the partitions; bpW is replicated.
template <int R>
__global__ __launch_bounds__(1024) void k_control(const double *__restrict__ bpW, const double *__restrict__ bpH, int nb,
                                                  const double *__restrict__ epart, int64_t nepart, double lgx, int r,
                                                  double n, double m_global, LoopCtl *ctl, double *__restrict__ history,
                                                  double *__restrict__ out_host, const double *__restrict__ tail_in,
                                                  const double *__restrict__ small_in)
{
    __shared__ double sW[R + 2], sT[R + 4];
    __shared__ double sm[1024];
    const int stopped = ctl->stop;               // tested below, once the loads of the reductions are in flight too
    double data;
    if (tail_in) {
        if (threadIdx.x < R + 2) sT[threadIdx.x] = tail_in[threadIdx.x];
        data = small_in[0]; lgx = small_in[1];
        bp_colsums(bpW, nb, R + 2, sW, 1024);
        if (stopped) return;
        __syncthreads();
    } else {
        double part = 0.0;                       // this thread's share of the evidence partials
        for (int64_t q = threadIdx.x; q < nepart; q += 1024) part += epart[q];
        bp_colsums2(bpW, bpH, nb, R + 2, sW, sT, 1024);
        if (stopped) return;
        data = block_sum(part, sm);
    }
    const int lane = threadIdx.x;
    if (lane > 1) return;                        // lanes 0 and 1 go on (the two Newton recurrences of hyper_update)
    double cross = 0.0, sew = 0.0, seh = 0.0;
    for (int k = 0; k < r; k++) { cross += sW[k] * sT[k]; sew += sW[k]; seh += sT[k]; }
    const double U = -cross - data - lgx + sW[R] + sT[R];
    const double lkh = U / (n * m_global);
    double st[4] = {sW[R + 1] / (n * r), sT[R + 1] / (m_global * r), sew / (n * r), seh / (m_global * r)};
    const int it = ctl->it + 1;
    int reason = 0;
    if (it > ctl->n0 && it % ctl->dn == 0) {
        if (dev_hyper_update_pair(ctl->flags, st, ctl->hyper, lane)) reason = 3;
    }
    if (lane != 0) return;
    const double lk0 = ctl->lk0;
    if (!reason) {
        if (lkh != lkh) reason = 1;
        else if (it > 1 && it > ctl->n0 && lkh >= lk0 && fabs(1.0 - lkh / lk0) < ctl->tol) reason = 2;
        else { ctl->lk0 = lkh; if (it >= ctl->max_it) reason = 4; }
    }
    ctl->it = it; ctl->lkh = lkh;
    for (int q = 0; q < 4; q++) ctl->stats[q] = st[q];
    if (history) {
        double *h = history + (size_t)(it - 1) * 9;
        h[0] = lkh;
        for (int q = 0; q < 4; q++) { h[1 + q] = st[q]; h[5 + q] = ctl->hyper[q]; }
    }
    if (reason) { ctl->reason = reason; ctl->stop = 1; }
    out_host[0] = lkh;
    for (int q = 0; q < 4; q++) { out_host[1 + q] = st[q]; out_host[8 + q] = ctl->hyper[q]; }
    out_host[12] = ctl->lk0;
    out_host[5] = (double)it;
    __threadfence_system();
    reinterpret_cast<volatile double *>(out_host)[6] = (double)reason;
    reinterpret_cast<volatile double *>(out_host)[7] = (double)it;
}

// Loads the control block of a device-driven loop (stream-ordered, no host copy to wait for).
__global__ void k_ctl_init(LoopCtl *ctl, const LoopCtl v) { *ctl = v; }

// Device-side evaluation of the special functions, for tests (tests/test_gpu_special.py).
__global__ void k_test_special(int kind, int64_t n, const double *__restrict__ x, double *__restrict__ y,
                               const LogTabEntry *__restrict__ tab)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double psi, lg;
    switch (kind) {
        case 0: y[i] = dev_log(x[i]); break;
        case 1: dev_psi_lgamma(x[i], &psi, &lg); y[i] = psi; break;
        case 2: dev_psi_lgamma(x[i], &psi, &lg); y[i] = lg; break;
        case 3: y[i] = dev_div(1.0, x[i]); break;
        case 4: y[i] = sp_rcp_seed(x[i]); break;             // raw v_rcp_f64
        case 5: y[i] = (double)__builtin_amdgcn_rcpf((float)x[i]); break;   // raw v_rcp_f32
        default: y[i] = dev_log_tab(x[i], tab); break;
    }
}

}  // namespace vbnmf
