// host.cpp -- host side of libvbnmf_hip.so: error plumbing, ingestion of X into the
// canonical CSC copy, and the builder of the tiled device layout.  No device code here.
//
// Reference behaviour this replaces: vb_iterate hands `as.matrix(bundle$mat)` to the
// native step on EVERY iteration (reference R/bayesian.R:339) and Rcpp copies it again
// into an Eigen::MatrixXd (reference src/RcppExports.cpp:15).  Here X is ingested once.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <sched.h>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

#include <chrono>
#include <system_error>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <numeric>

namespace vbnmf {

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const char *last_error_cstr() { return g_err.c_str(); }

// ------------------------------------------------------------------ threads
static std::atomic<int> g_host_threads_override{0};         // vbnmf_set_host_threads (0: the default rule below)
void set_host_threads_override(int n) { g_host_threads_override.store(n > 0 ? std::min(n, 1024) : 0, std::memory_order_relaxed); }

int host_threads()
{
    const int forced = g_host_threads_override.load(std::memory_order_relaxed);
    if (forced > 0) return forced;
    static int n = [] {
        if (const char *s = getenv("VBNMF_HOST_THREADS")) {
            int v = atoi(s);
            if (v > 0) return v;
        }
        unsigned hc = std::thread::hardware_concurrency();
        int v = hc ? (int)hc : 1;
        cpu_set_t set;                                   // a rank pinned to a subset of the cores uses only those
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) v = std::min(v, CPU_COUNT(&set));
        return std::min(v, 32);
    }();
    return n;
}

// Fork-join over [0, count): every call starts and joins its own threads (~25 us apiece on the GPU box's host).  Round 5 built
// the alternative -- a pool of workers that outlives the call, the caller helping -- and measured it on the headline matrix's
// set-up (profiles/r05_setup_pool_ab.txt, r05_transpose_ab2.txt): the dispatch is cheaper, but the heavy phases run SLOWER on
// woken workers than on fresh threads (the fill of the gene side 0.07-0.09 s against 0.04-0.05 s, engine creation 0.30-0.36 s
// against 0.25-0.28 s): a new thread is placed on the idlest core of a 256-CPU host, a woken one near where it last ran or near
// its waker, and a phase of 40 ms is over before the balancer has spread them.  So: threads per call, and callers whose work is
// small ask for few of them (the state conversions of set_state / get_state, the cache key of the stateless entries).
static thread_local int tl_thread_share = 1;          // this host thread's calls use host_threads() / share threads by default
void set_thread_share(int share) { tl_thread_share = share > 1 ? share : 1; }
static int default_threads() { return std::max(1, host_threads() / tl_thread_share); }

void parallel_for(int64_t count, const std::function<void(int64_t, int64_t, int)> &fn, int max_threads)
{
    if (count <= 0) return;
    int nt = max_threads > 0 ? max_threads : default_threads();
    if ((int64_t)nt > count) nt = (int)count;
    if (nt <= 1) { fn(0, count, 0); return; }
    // An exception leaving a std::thread ends the process (std::terminate): a worker that runs out of memory hands its
    // exception to the calling thread instead, which rethrows it once every worker has been joined -- the C ABI's entry
    // points then turn it into VBNMF_ERR_OOM like any other allocation failure.
    std::exception_ptr first;
    std::mutex first_mu;
    auto guarded = [&](int64_t b, int64_t e, int t) {
        try {
            fn(b, e, t);
        } catch (...) {
            std::lock_guard<std::mutex> g(first_mu);
            if (!first) first = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 1; t < nt; t++) {
        int64_t b = count * t / nt, e = count * (t + 1) / nt;
        try {
            th.emplace_back([&guarded, b, e, t] { guarded(b, e, t); });
        } catch (const std::system_error &) {            // thread limit reached: this piece runs here
            guarded(b, e, t);
        }
    }
    guarded(0, count / nt, 0);                           // (the caller takes the first piece instead of sleeping)
    for (auto &x : th) x.join();
    if (first) std::rethrow_exception(first);
}

// ------------------------------------------------------------------ ingestion
static void finish_matrix(Matrix &X)
{
    X.nnz = X.colptr[X.m];
    // one pass over the values on all host threads (it was a serial 5e7-element loop: 80 ms of the headline's ingestion)
    const int T = host_threads();
    std::vector<char> ints_t(T, 1);
    std::vector<double> mx_t(T, 0.0);
    parallel_for(X.nnz, [&](int64_t b, int64_t e, int tid) {
        bool ok = true;
        double m = 0.0;
        for (int64_t q = b; q < e; q++) {
            const double v = X.val[q];
            ok = ok && v >= 1.0 && v < 2147483648.0 && v == std::floor(v);
            m = std::max(m, v);
        }
        ints_t[tid] = ok ? 1 : 0; mx_t[tid] = m;
    }, T);
    bool ints = true;
    double mx = 0.0;
    for (int t = 0; t < T; t++) { ints = ints && ints_t[t]; mx = std::max(mx, mx_t[t]); }
    X.counts_int = ints;
    X.max_val = mx;
    X.counts_u16 = ints && mx <= kPackedCountMax;
}

template <class VI, class VD>
static void transpose_compressed(int64_t nouter, int64_t ninner, const int64_t *ptr, const int32_t *idx, const double *val,
                                 int32_t idx_offset, std::vector<int64_t> &tptr, VI &tidx, VD &tval, const int32_t *perm);

const RowMajor &Matrix::row_major() const
{
    std::call_once(rm_cache->once, [&] {
        const std::vector<int32_t> &perm = cell_order();
        transpose_compressed(m, n, colptr.data(), row.data(), val.data(), 0, rm_cache->rm.ptr, rm_cache->rm.idx, rm_cache->rm.val,
                             perm.empty() ? nullptr : perm.data());
    });
    return rm_cache->rm;
}

static int matrix_from_dense(int64_t n, int64_t m, const double *A, Matrix &X)
{
    X.n = n; X.m = m;
    X.colptr.assign(m + 1, 0);
    parallel_for(m, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            const double *c = A + (size_t)j * n;
            int64_t k = 0;
            for (int64_t i = 0; i < n; i++) k += (c[i] != 0.0);
            X.colptr[j + 1] = k;
        }
    });
    for (int64_t j = 0; j < m; j++) X.colptr[j + 1] += X.colptr[j];
    int64_t nnz = X.colptr[m];
    X.row.resize(nnz);
    X.val.resize(nnz);
    parallel_for(m, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            const double *c = A + (size_t)j * n;
            int64_t o = X.colptr[j];
            for (int64_t i = 0; i < n; i++)
                if (c[i] != 0.0) { X.row[o] = (int32_t)i; X.val[o] = c[i]; o++; }
        }
    });
    finish_matrix(X);
    return VBNMF_OK;
}

// Compressed input with `nouter` outer vectors of inner indices < ninner.  Produces the
// canonical form in the same orientation (inner ascending, duplicates summed, zeros dropped).
static int canonicalise(int64_t nouter, int64_t ninner, const int32_t *p, const int32_t *idx, const double *x,
                        std::vector<int64_t> &optr, std::vector<int32_t> &oidx, std::vector<double> &oval)
{
    if (p[0] != 0) return fail(VBNMF_ERR_BAD_ARG, "pointer array must start at 0");
    for (int64_t j = 0; j < nouter; j++)
        if (p[j + 1] < p[j]) return fail(VBNMF_ERR_BAD_ARG, "pointer array is not non-decreasing at %lld", (long long)j);
    int64_t nin = p[nouter];
    {
        std::atomic<int64_t> bad{-1};                       // the first offending position any thread saw (smallest wins below)
        parallel_for(nin, [&](int64_t b, int64_t e, int) {
            for (int64_t q = b; q < e; q++)
                if (idx[q] < 0 || idx[q] >= ninner) {
                    int64_t cur = bad.load();
                    while ((cur < 0 || q < cur) && !bad.compare_exchange_weak(cur, q)) {}
                    break;
                }
        });
        const int64_t e = bad.load();
        if (e >= 0) return fail(VBNMF_ERR_BAD_ARG, "index %d at position %lld is outside [0, %lld)", idx[e], (long long)e, (long long)ninner);
    }
    optr.assign(nouter + 1, 0);
    std::vector<int64_t> kept(nouter, 0);
    // pass 1: per outer vector, sort a scratch copy and count surviving entries
    std::vector<int32_t> sidx(nin);
    std::vector<double> sval(nin);
    parallel_for(nouter, [&](int64_t b, int64_t e, int) {
        std::vector<std::pair<int32_t, double>> tmp;
        for (int64_t j = b; j < e; j++) {
            int64_t s = p[j], t = p[j + 1];
            bool sorted = true;
            for (int64_t q = s + 1; q < t; q++)
                if (idx[q] <= idx[q - 1]) { sorted = false; break; }
            int64_t o = s;
            if (sorted) {
                for (int64_t q = s; q < t; q++)
                    if (x[q] != 0.0) { sidx[o] = idx[q]; sval[o] = x[q]; o++; }
            } else {
                tmp.clear();
                for (int64_t q = s; q < t; q++) tmp.emplace_back(idx[q], x[q]);
                std::stable_sort(tmp.begin(), tmp.end(),
                                 [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &c) { return a.first < c.first; });
                size_t q = 0;
                while (q < tmp.size()) {
                    int32_t id = tmp[q].first;
                    double v = 0.0;
                    while (q < tmp.size() && tmp[q].first == id) { v += tmp[q].second; q++; }
                    if (v != 0.0) { sidx[o] = id; sval[o] = v; o++; }
                }
            }
            kept[j] = o - s;
        }
    });
    for (int64_t j = 0; j < nouter; j++) optr[j + 1] = optr[j] + kept[j];
    oidx.resize(optr[nouter]);
    oval.resize(optr[nouter]);
    parallel_for(nouter, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            int64_t s = p[j], o = optr[j];
            for (int64_t q = 0; q < kept[j]; q++) { oidx[o + q] = sidx[s + q]; oval[o + q] = sval[s + q]; }
        }
    });
    return VBNMF_OK;
}

int matrix_from_csc(int64_t n, int64_t m, const int32_t *p, const int32_t *i, const double *x, Matrix &X)
{
    X.n = n; X.m = m;
    int rc = canonicalise(m, n, p, i, x, X.colptr, X.row, X.val);
    if (rc) return rc;
    finish_matrix(X);
    return VBNMF_OK;
}

// Transpose a canonical compressed matrix (nouter x ninner) into the other orientation (inner indices of the
// result ascending).  The OUTPUT is what the threads divide: the inner indices are cut into ranges (a few hundred rows of the
// result each), and since every outer vector holds its inner indices in ascending order, the stretch of it that falls into a
// range is found by bisection (a table of nouter x (ranges + 1) offsets, one pass).  A thread then takes a range: it counts the
// range's entries per inner index, scans, and scatters them -- outer vectors in order, so the result is ascending -- with all its
// write cursors (a few hundred) and its output stretch in its own cache.  The earlier form divided the INPUT (one chunk of outer
// vectors per thread, a counter per (chunk, inner index), every thread scattering into all 20 000 rows of the result): 0.117 s
// for the headline matrix's 4.9e7 entries on the GPU box's host, most of it cache misses of the scatter; this one 0.06-0.08 s
// (profiles/r05_transpose_ab2.txt).  Round 5 also measured a two-level form (entries first into <= 256 buckets as 16-byte records,
// then a stable pass inside every bucket): 0.27 s -- the record buffer's fresh 800 MB cost more than the scattered cursors.
// The result does not depend on the thread count (each range's content and order are fixed by the input alone).
// perm (optional, nouter entries): outer vector p of the result's numbering is input vector perm[p].  VI / VD: vectors of
// int32_t / double (BigVec where the caller can take it: the 600 MB of the result are then first touched by the threads that
// write them instead of being zero-filled by the caller's one).
template <class VI, class VD>
static void transpose_compressed(int64_t nouter, int64_t ninner, const int64_t *ptr, const int32_t *idx, const double *val,
                                 int32_t idx_offset, std::vector<int64_t> &tptr, VI &tidx, VD &tval, const int32_t *perm)
{
    const int64_t s = ptr[0], t = ptr[nouter];
    const auto t_0 = std::chrono::steady_clock::now();
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(default_threads(), (t - s) / (1 << 18) + 1));
    auto src = [&](int64_t j) { return perm ? (int64_t)perm[j] : j; };          // input vector stored at position j
    tptr.assign(ninner + 1, 0);
    tidx.resize(t - s);
    tval.resize(t - s);
    if (ninner <= 0) return;
    // ranges of inner indices: a few per thread (they differ in entries; handed out through a counter), one when serial
    int64_t G = T == 1 ? 1 : std::min<int64_t>(ninner, 4 * (int64_t)T);
    while (G > 1 && nouter * (G + 1) > ((int64_t)1 << 26)) G /= 2;                 // (the table of offsets below: at most 256 MB)
    auto lo_of = [&](int64_t g) { return ninner * g / G; };
    // split[j * (G + 1) + g]: offset inside outer vector j (position order) of its first entry with inner index >= lo_of(g)
    std::vector<uint32_t> split((size_t)nouter * (size_t)(G + 1));
    std::vector<int64_t> tot_t((size_t)T * (size_t)G, 0);                       // entries per range, by thread
    parallel_for(nouter, [&](int64_t b, int64_t e, int tid) {
        int64_t *tot = &tot_t[(size_t)tid * (size_t)G];
        for (int64_t j = b; j < e; j++) {
            const int64_t c = src(j);
            const int32_t *first = idx + ptr[c], *last = idx + ptr[c + 1];
            uint32_t *sp = &split[(size_t)j * (size_t)(G + 1)];
            const int32_t *at = first;
            sp[0] = 0;
            for (int64_t g = 1; g < G; g++) {
                at = std::lower_bound(at, last, (int32_t)lo_of(g));
                sp[g] = (uint32_t)(at - first);
                tot[g - 1] += (int64_t)sp[g] - (int64_t)sp[g - 1];
            }
            sp[G] = (uint32_t)(last - first);
            tot[G - 1] += (int64_t)sp[G] - (int64_t)sp[G - 1];
        }
    }, T);
    static const bool tt = getenv("VBNMF_BUILD_TIMES") != nullptr;
    auto t_a = std::chrono::steady_clock::now();
    std::vector<int64_t> base((size_t)G + 1, 0);
    for (int64_t g = 0; g < G; g++) {
        int64_t k = 0;
        for (int c = 0; c < T; c++) k += tot_t[(size_t)c * (size_t)G + (size_t)g];
        base[g + 1] = base[g] + k;
    }
    std::atomic<int64_t> next{0};
    parallel_for(T, [&](int64_t, int64_t, int) {
        std::vector<int64_t> cur;
        for (;;) {
            const int64_t g = next.fetch_add(1);
            if (g >= G) break;
            const int64_t lo = lo_of(g), hi = lo_of(g + 1);
            cur.assign((size_t)(hi - lo) + 1, 0);
            for (int64_t j = 0; j < nouter; j++) {
                const uint32_t *sp = &split[(size_t)j * (size_t)(G + 1) + (size_t)g];
                const int32_t *q = idx + ptr[src(j)];
                for (uint32_t u = sp[0]; u < sp[1]; u++) cur[(size_t)(q[u] - lo) + 1]++;
            }
            int64_t run = base[g];
            for (int64_t i = 0; i < hi - lo; i++) { const int64_t k = cur[(size_t)i + 1]; tptr[lo + i] = run; cur[(size_t)i] = run; run += k; }
            for (int64_t j = 0; j < nouter; j++) {
                const uint32_t *sp = &split[(size_t)j * (size_t)(G + 1) + (size_t)g];
                const int64_t c0 = ptr[src(j)];
                const int32_t *q = idx + c0;
                const double *v = val + c0;
                for (uint32_t u = sp[0]; u < sp[1]; u++) {
                    const int64_t o = cur[(size_t)(q[u] - lo)]++;
                    tidx[o] = (int32_t)(j - idx_offset);
                    tval[o] = v[u];
                }
            }
        }
    }, T);
    tptr[ninner] = t - s;
    if (tt) fprintf(stderr, "  transpose: T %d, G %lld, after the split table %.4f s, ranges %.4f s\n", T, (long long)G,
                    std::chrono::duration<double>(t_a - t_0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count());
}

static int matrix_from_csr(int64_t n, int64_t m, const int32_t *p, const int32_t *j, const double *x, Matrix &X)
{
    std::vector<int64_t> rptr;
    std::vector<int32_t> ridx;
    std::vector<double> rval;
    int rc = canonicalise(n, m, p, j, x, rptr, ridx, rval);
    if (rc) return rc;
    X.n = n; X.m = m;
    transpose_compressed(n, m, rptr.data(), ridx.data(), rval.data(), 0, X.colptr, X.row, X.val, nullptr);
    finish_matrix(X);
    return VBNMF_OK;
}

// sum_ij lgamma(X_ij + 1) over stored entries (absent entries give lgamma(1) = 0): the
// iteration-invariant part of reference src/vbnmf_update.cpp:80-81.  Per-column sums are
// formed independently and then added in column order, so the value does not depend on the
// host thread count.
double sum_lgamma_x1(const Matrix &X, int64_t cb, int64_t ce)
{
    std::vector<double> table;
    if (X.counts_u16) {
        table.resize((size_t)X.max_val + 1);                // (the matrix's maximum bounds every column range's)
        for (size_t c = 0; c < table.size(); c++) table[c] = std::lgamma((double)c + 1.0);
    }
    std::vector<double> colsum(ce - cb, 0.0);
    parallel_for(ce - cb, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            double s = 0.0;
            for (int64_t q = X.colptr[cb + j]; q < X.colptr[cb + j + 1]; q++)
                s += X.counts_u16 ? table[(size_t)X.val[q]] : std::lgamma(X.val[q] + 1.0);
            colsum[j] = s;
        }
    });
    double s = 0.0;
    for (double v : colsum) s += v;
    return s;
}

// sum over stored entries of -x log x + x (reference R/factorize.R:46-47), columns [cb, ce), fixed order.
double sum_xlogx(const Matrix &X, int64_t cb, int64_t ce)
{
    std::vector<double> table;
    if (X.counts_u16) {
        table.resize((size_t)X.max_val + 1);
        table[0] = 0.0;
        for (size_t c = 1; c < table.size(); c++) table[c] = -(double)c * std::log((double)c) + (double)c;
    }
    std::vector<double> colsum(ce - cb, 0.0);
    parallel_for(ce - cb, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            double s = 0.0;
            for (int64_t q = X.colptr[cb + j]; q < X.colptr[cb + j + 1]; q++) {
                const double v = X.val[q];
                s += X.counts_u16 ? table[(size_t)v] : (v > 0.0 ? -v * std::log(v) + v : 0.0);
            }
            colsum[j] = s;
        }
    });
    double s = 0.0;
    for (double v : colsum) s += v;
    return s;
}

// ------------------------------------------------------------------ layout
static int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    return atoi(s);
}

LayoutParams default_layout_params(int64_t n_major, int64_t n_minor, int R, int n_wg, int64_t nnz)
{
    LayoutParams lp;
    // One minor block of the gathered factor must fit the workgroup's LDS: block_width * R * 8 bytes.
    int lds_kb = env_int("VBNMF_LDS_KB", 160);
    if (lds_kb < 8) lds_kb = 8;
    if (lds_kb > 160) lds_kb = 160;
    int64_t cmax = ((int64_t)lds_kb * 1024 - kLdsReserveBytes) / lds_row_bytes(R);
    if (n_wg <= 0) n_wg = env_int("VBNMF_NWG", 256);
    if (n_wg < 1) n_wg = 1;
    // Longest task.  A lane walks its task serially (~0.15 us per entry when its wave is alone on a SIMD), so on a
    // small matrix 256-entry tasks leave a handful of waves running for 40 us while the rest of the chip idles:
    // tasks are cut short enough that every wave of every workgroup can have work, up to the 256 that the
    // headline size wants (shorter tasks there only add partial rows).  VBNMF_MAX_LEN overrides.
    int ml = env_int("VBNMF_MAX_LEN", 0);
    if (ml <= 0) {
        const int64_t waves = (int64_t)n_wg * (sweep_threads(R) / kLanes);
        // measured on 1 000 x 450, 2 000 x 5 000, 5 000 x 10 000 and the headline matrix: best near two entries per lane
        // of every wave, not below 16
        ml = nnz > 0 ? (int)std::max<int64_t>(16, std::min<int64_t>(256, 2 * nnz / (kLanes * waves) + 1)) : 256;
    }
    if (ml < kWidthQuantum) ml = kWidthQuantum;
    ml = (ml + kWidthQuantum - 1) / kWidthQuantum * kWidthQuantum;
    // Dense-ish matrices: a (major, block) pair much longer than the longest task is cut into several tasks anyway, so a
    // narrower block costs no extra task and stages less.  About three tasks per pair at the matrix's mean density
    // (2 000 x 10 000, 75 % stored, rank 5: 78.1 us per step with 160 KB blocks, 76.7 with 80, 75.4 with 32 -- this rule --,
    // 81.8 with 16); never below 256 rows; at 5 % density the LDS capacity is the tighter bound by far.
    if (nnz > 0 && n_major > 0 && n_minor > 0 && env_int("VBNMF_LDS_KB", 0) == 0) {
        const double density = (double)nnz / ((double)n_major * (double)n_minor);
        const int64_t want = (int64_t)std::max(256.0, 3.0 * (double)ml / std::max(density, 1e-9));
        if (want < cmax) cmax = want;
    }
    cmax &= ~(int64_t)7;
    if (cmax > 65528) cmax = 65528;            // local minor index is 16 bits
    if (cmax < 8) cmax = 8;
    int64_t nb = (n_minor + cmax - 1) / cmax;
    int64_t c = (n_minor + nb - 1) / nb;       // equal-width blocks instead of a short last one
    c = (c + 7) & ~(int64_t)7;
    if (c > cmax) c = cmax;
    lp.block_width = (int32_t)c;
    lp.block_cap = (int32_t)cmax;
    lp.n_wg = n_wg;
    lp.max_len = ml;
    lp.row_slots = lds_row_bytes(R) / 16;
    return lp;
}

int build_layout(const Matrix &X, int64_t cb, int64_t ce, int side, const LayoutParams &lp, const std::vector<int32_t> *perm_in, Layout &L,
                 LayoutSink *sink)
{
    const int32_t *perm = (perm_in && !perm_in->empty()) ? perm_in->data() : nullptr;
    if (perm && (int64_t)perm_in->size() != ce - cb) return fail(VBNMF_ERR_BAD_ARG, "cell order has %lld entries for %lld cells", (long long)perm_in->size(), (long long)(ce - cb));
    if (cb < 0 || ce > X.m || cb >= ce) return fail(VBNMF_ERR_BAD_ARG, "column range [%lld, %lld) is outside the matrix", (long long)cb, (long long)ce);
    // (max_len <= 0x7FF8: the two leading-stretch lengths of a slice share one int32, 15 + 16 bits -- slice_fast below)
    if (lp.block_width <= 0 || lp.block_width > 65536 || lp.max_len <= 0 || lp.max_len > 0x7FF8 || lp.max_len % kWidthQuantum || lp.n_wg <= 0 || lp.row_slots <= 0 ||
        (int64_t)std::max(lp.block_width, lp.block_cap) * lp.row_slots > (int64_t)(kPackedOffsetMask >> 4) + 1)
        return fail(VBNMF_ERR_BAD_ARG, "bad layout parameters");

    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (getenv("VBNMF_BUILD_TIMES")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  layout side %d %-12s %.3f s\n", side, what, std::chrono::duration<double>(t - T0).count()); T0 = t; } };
    // major-compressed view of X[:, cb:ce)
    std::vector<int64_t> tptr;
    BigVec<int32_t> tidx;
    BigVec<double> tval;
    // entries of major M: [pb[M], pe[M]) of idx / val.  Contiguous majors: pb = ptr, pe = ptr + 1; the cell side under a
    // renumbering of the cells walks the columns in the new order through two index arrays instead (no copy of X).
    const int64_t *pb, *pe;
    const int32_t *idx;
    const double *val;
    std::vector<int64_t> obeg, oend;
    if (side == 1) {
        L.n_major = ce - cb; L.n_minor = X.n;
        idx = X.row.data(); val = X.val.data();
        if (perm) {
            obeg.resize(ce - cb); oend.resize(ce - cb);
            for (int64_t p = 0; p < ce - cb; p++) { obeg[p] = X.colptr[cb + perm[p]]; oend[p] = X.colptr[cb + perm[p] + 1]; }
            pb = obeg.data(); pe = oend.data();
        } else {
            pb = X.colptr.data() + cb; pe = pb + 1;
        }
    } else {
        L.n_major = X.n; L.n_minor = ce - cb;
        if (cb == 0 && ce == X.m) {
            const RowMajor &Rm = X.row_major();               // built once per matrix (in the matrix's cell order), shared by every engine on it
            pb = Rm.ptr.data(); idx = Rm.idx.data(); val = Rm.val.data();
        } else {
            transpose_compressed(ce - cb, X.n, X.colptr.data() + cb, X.row.data(), X.val.data(), 0, tptr, tidx, tval, perm);
            pb = tptr.data(); idx = tidx.data(); val = tval.data();
        }
        pe = pb + 1;
    }
    if (perm) L.cell_perm.assign(perm, perm + (ce - cb));
    lap("transpose");
    L.side = side;
    L.wide = !X.counts_int;
    L.nnz = X.colptr[ce] - X.colptr[cb];                 // stored entries of X (before any splitting below)
    // integer counts above the packed range: the entry is stored as ceil(x / kPackedCountMax) entries of the same minor
    std::vector<int64_t> xptr;
    std::vector<int32_t> xidx;
    std::vector<double> xval;
    if (!L.wide && X.max_val > kPackedCountMax) {
        const int64_t nm = L.n_major;
        xptr.assign(nm + 1, 0);
        for (int64_t M = 0; M < nm; M++) {
            int64_t c = 0;
            for (int64_t q = pb[M]; q < pe[M]; q++) c += (int64_t)std::ceil(val[q] / kPackedCountMax);
            xptr[M + 1] = xptr[M] + c;
        }
        xidx.resize(xptr[nm]); xval.resize(xptr[nm]);
        parallel_for(nm, [&](int64_t b, int64_t e, int) {
            for (int64_t M = b; M < e; M++) {
                int64_t o = xptr[M];
                for (int64_t q = pb[M]; q < pe[M]; q++) {
                    double left = val[q];
                    while (left > 0.0) {
                        const double piece = std::min(left, kPackedCountMax);
                        xidx[o] = idx[q]; xval[o] = piece; o++;
                        left -= piece;
                    }
                }
            }
        });
        pb = xptr.data(); pe = pb + 1; idx = xidx.data(); val = xval.data();
    }
    L.max_len = lp.max_len;
    L.n_wg = lp.n_wg;
    L.row_slots = lp.row_slots;
    const int64_t nmaj = L.n_major;
    // Minor blocks.  Whole workgroups are handed to blocks (a workgroup stages ONE block per side), so a block whose
    // cost is 9.8 workgroups' worth gets 10 or 9 of them -- and in the second case each of its workgroups carries 9 %
    // more than the rest (measured on the headline matrix, gene side: 26 equal blocks, 22 with 10 workgroups and 4
    // with 9: modelled cost max / mean 1.13, and the slowest workgroups took 100 us against a mean of 88).  The block
    // boundaries are therefore put where the cumulative entry count reaches a whole number of workgroup quotas:
    // every block is worth an integer G_b of them (G_b as equal as possible), no wider than the LDS allows.
    // With more blocks than workgroups (huge matrices) the blocks stay equal and are bin-packed below.
    const int32_t wmax = lp.block_cap > 0 ? lp.block_cap : lp.block_width;
    std::vector<int64_t> bstart;
    {
        std::vector<int64_t> mcount(L.n_minor + 1, 0);                     // entries per minor -> prefix sums
        {
            const int T = host_threads();
            std::vector<std::vector<int64_t>> part(T);
            parallel_for(nmaj, [&](int64_t b, int64_t e, int tid) {
                std::vector<int64_t> &c = part[tid];
                c.assign(L.n_minor, 0);
                for (int64_t M = b; M < e; M++)
                    for (int64_t q = pb[M]; q < pe[M]; q++) c[idx[q]]++;
            }, T);
            for (const auto &c : part)
                if (!c.empty()) for (int64_t j = 0; j < L.n_minor; j++) mcount[j + 1] += c[j];
        }
        for (int64_t j = 0; j < L.n_minor; j++) mcount[j + 1] += mcount[j];
        const int64_t total = mcount[L.n_minor];
        int64_t nb = (L.n_minor + wmax - 1) / wmax;
        const bool proportional = env_int("VBNMF_EQUAL_BLOCKS", 0) == 0 && total > 0;
        for (; proportional && nb <= lp.n_wg; nb++) {
            std::vector<int64_t> cand(nb + 1, 0);
            bool ok = true;
            int64_t gsum = 0;
            for (int64_t b = 0; b < nb && ok; b++) {
                gsum += lp.n_wg / nb + (b < lp.n_wg % nb ? 1 : 0);              // G_b: as equal as possible
                int64_t end = L.n_minor;
                if (b + 1 < nb) {
                    const double target = (double)total * (double)gsum / (double)lp.n_wg;
                    end = std::lower_bound(mcount.begin(), mcount.end(), (int64_t)std::llround(target)) - mcount.begin();
                    end = std::min<int64_t>(std::max<int64_t>(end, cand[b] + 1), L.n_minor - (nb - 1 - b));
                }
                cand[b + 1] = end;
                ok = end - cand[b] <= wmax;
            }
            if (ok) { bstart.swap(cand); break; }
        }
        if (bstart.empty()) {                                                   // equal blocks
            nb = (L.n_minor + lp.block_width - 1) / lp.block_width;
            bstart.resize(nb + 1);
            for (int64_t b = 0; b <= nb; b++) bstart[b] = std::min<int64_t>(L.n_minor, b * (int64_t)lp.block_width);
        }
    }
    const int32_t nblk = (int32_t)bstart.size() - 1;
    L.n_blocks = nblk;
    L.block_start.assign(bstart.begin(), bstart.end());
    int32_t widest = 0;
    for (int32_t b = 0; b < nblk; b++) widest = std::max<int32_t>(widest, (int32_t)(bstart[b + 1] - bstart[b]));
    L.block_width = widest;                                                     // what the LDS image is sized for
    lap("blocks");

    // bpos[major][b] = position of the major's first entry whose minor is in block >= b
    std::vector<int64_t> bpos((size_t)nmaj * (nblk + 1));
    parallel_for(nmaj, [&](int64_t b, int64_t e, int) {
        for (int64_t M = b; M < e; M++) {
            int64_t q = pb[M], t = pe[M];
            int64_t *bp = &bpos[(size_t)M * (nblk + 1)];
            for (int32_t blk = 0; blk <= nblk; blk++) {
                int64_t lim = bstart[blk];
                while (q < t && idx[q] < lim) q++;
                bp[blk] = q;
            }
        }
    });

    lap("bpos");
    // tasks per block: (major, block) runs longer than max_len are cut in near-equal pieces
    // n1 = the task's entries of value exactly 1: they are placed first in the task, so that the sweep can run the
    // leading trips of a slice -- as far as EVERY lane still sits on such entries -- through a shorter loop (no
    // count conversion, and on the gene side a running product in place of a logarithm per entry, kernels.h).
    // Tasks are therefore grouped by padded length first and, within a length class, by n1: the 64 tasks of a slice
    // then agree on how long that leading stretch is.
    struct Task { uint32_t major; int32_t len; int64_t pos; int32_t n1, n2; };      // n2: entries of value exactly 2 (placed after the ones)
    const bool fast_ones = !L.wide && env_int("VBNMF_NO_FAST_ONES", 0) == 0;
    std::vector<std::vector<Task>> btasks(nblk);
    parallel_for(nblk, [&](int64_t b0, int64_t b1, int) {
        for (int64_t blk = b0; blk < b1; blk++) {
            std::vector<Task> &T = btasks[blk];
            for (int64_t M = 0; M < nmaj; M++) {
                const int64_t *bp = &bpos[(size_t)M * (nblk + 1)];
                int64_t q0 = bp[blk], cnt = bp[blk + 1] - q0;
                if (cnt <= 0) continue;
                int64_t pieces = (cnt + lp.max_len - 1) / lp.max_len;
                for (int64_t pc = 0; pc < pieces; pc++) {
                    int64_t s = cnt * pc / pieces, t = cnt * (pc + 1) / pieces;
                    int32_t n1 = 0, n2 = 0;
                    if (fast_ones) for (int64_t q = q0 + s; q < q0 + t; q++) { n1 += (val[q] == 1.0); n2 += (val[q] == 2.0); }
                    T.push_back({(uint32_t)M, (int32_t)(t - s), q0 + s, n1, n2});
                }
            }
            auto padded = [](int32_t len) { return (len + kWidthQuantum - 1) / kWidthQuantum; };
            std::stable_sort(T.begin(), T.end(), [&](const Task &a, const Task &c2) {
                const int32_t pa = padded(a.len), pc2 = padded(c2.len);
                if (pa != pc2) return pa > pc2;
                // inside a length class by the number of ones -- descending in even classes, ascending in odd ones, so
                // that the slice that straddles two classes joins tasks with ALIKE counts (its leading stretch is the
                // minimum over its lanes): gene side of the headline matrix, stretch 48.0 -> 53.9 % of the slots
                return (pa & 1) ? a.n1 < c2.n1 : a.n1 > c2.n1;
            });
        }
    });

    lap("tasks");
    // slices: 64 consecutive tasks of a block; blocks in index order
    std::vector<int64_t> bslice0(nblk + 1, 0);
    for (int32_t blk = 0; blk < nblk; blk++)
        bslice0[blk + 1] = bslice0[blk] + ((int64_t)btasks[blk].size() + kLanes - 1) / kLanes;
    L.n_slices = bslice0[nblk];
    if (L.n_slices > 0x7FFFFFF0LL / kLanes) return fail(VBNMF_ERR_BAD_ARG, "too many tasks for 32-bit task ids");
    L.task_major.assign((size_t)L.n_slices * kLanes, kIdleLane);
    L.slice_width.assign(L.n_slices, 0);
    L.slice_off.assign(L.n_slices, 0);
    L.slice_block.assign(L.n_slices, 0);
    std::vector<int64_t> task_pos((size_t)L.n_slices * kLanes, 0);
    std::vector<int32_t> task_len((size_t)L.n_slices * kLanes, 0);
    std::vector<int32_t> task_n1((size_t)L.n_slices * kLanes, 0), task_n2((size_t)L.n_slices * kLanes, 0);
    L.slice_fast.assign(L.n_slices, 0);
    L.n_tasks = 0;
    for (int32_t blk = 0; blk < nblk; blk++) {
        const std::vector<Task> &T = btasks[blk];
        L.n_tasks += (int64_t)T.size();
        for (size_t q = 0; q < T.size(); q++) {
            size_t id = (size_t)bslice0[blk] * kLanes + q;
            L.task_major[id] = T[q].major; task_pos[id] = T[q].pos; task_len[id] = T[q].len; task_n1[id] = T[q].n1; task_n2[id] = T[q].n2;
        }
        for (int64_t s = bslice0[blk]; s < bslice0[blk + 1]; s++) {
            int32_t w = task_len[(size_t)s * kLanes];            // sorted by padded length: the first lane's is the largest
            L.slice_width[s] = (w + kWidthQuantum - 1) / kWidthQuantum * kWidthQuantum;
            L.slice_block[s] = blk;
            int32_t f = INT32_MAX;                               // leading entries that are ones in EVERY lane (idle lanes: none)
            int32_t f12 = INT32_MAX;                             // ... that are ones or twos in every lane (ones first, then twos)
            for (int l = 0; l < kLanes; l++) {
                f = std::min(f, task_n1[(size_t)s * kLanes + l]);
                f12 = std::min(f12, task_n1[(size_t)s * kLanes + l] + task_n2[(size_t)s * kLanes + l]);
            }
            const int32_t f1 = std::min<int32_t>(f / 8 * 8, 0x7FF8);   // whole loop trips (8 entries)
            const int32_t f2 = std::min<int32_t>(std::max(f1, f12 / 8 * 8), 0x7FF8);      // (sign bit of the word stays clear)
            L.slice_fast[s] = f1 | (f2 << 16);                   // low half: the stretch of ones; high half: of ones and twos
        }
    }
    btasks.clear();
    int64_t off = 0;
    for (int64_t s = 0; s < L.n_slices; s++) { L.slice_off[s] = off; off += (int64_t)L.slice_width[s] * kLanes; }
    L.n_slots = off;

    // inverse index: the tasks of each major in (block, position) order -- the fixed order in
    // which their partial statistics are summed (built below, once the slices have their final numbers)
    auto build_inverse = [&]() {
    L.inv_ptr.assign(nmaj + 1, 0);
    for (size_t id = 0; id < L.task_major.size(); id++)
        if (L.task_major[id] != kIdleLane) L.inv_ptr[L.task_major[id] + 1]++;
    for (int64_t M = 0; M < nmaj; M++) L.inv_ptr[M + 1] += L.inv_ptr[M];
    L.inv_task.assign(L.n_tasks, 0);
    {
        std::vector<std::pair<int64_t, uint32_t>> tmp;     // (position, id) per major
        std::vector<int32_t> cur(L.inv_ptr.begin(), L.inv_ptr.end() - 1);
        std::vector<int64_t> key(L.n_tasks);
        for (size_t id = 0; id < L.task_major.size(); id++) {
            uint32_t M = L.task_major[id];
            if (M == kIdleLane) continue;
            int32_t o = cur[M]++;
            L.inv_task[o] = (uint32_t)id; key[o] = task_pos[id];
        }
        parallel_for(nmaj, [&](int64_t b, int64_t e, int) {
            std::vector<std::pair<int64_t, uint32_t>> t2;
            for (int64_t M = b; M < e; M++) {
                int32_t s = L.inv_ptr[M], t = L.inv_ptr[M + 1];
                if (t - s < 2) continue;
                t2.clear();
                for (int32_t q = s; q < t; q++) t2.emplace_back(key[q], L.inv_task[q]);
                std::sort(t2.begin(), t2.end());
                for (int32_t q = s; q < t; q++) L.inv_task[q] = t2[q - s].second;
            }
        });
    }
    };

    lap("slices");
    // persistent workgroups.  Shares are block-aligned so a workgroup stages one block per side:
    // whole workgroups are apportioned to blocks in proportion to block cost (largest remainder);
    // a block's slices, sorted by width, are dealt to its workgroups in snake order (equal cost,
    // same mix of long and short slices); inside a share the waves pull the slices longest first at run
    // time (see below).  With more blocks than workgroups, whole blocks are bin-packed onto workgroups instead.
    {
        const double c0 = 10.0;                              // per-slice overhead in entry-equivalents
        // an entry of the leading stretch of ones costs the gene side (which carries the logarithm) ~0.6 and the cell
        // side ~0.9 of an ordinary entry (instruction counts of the two loops, kernels.h)
        const double fast_discount = side == 0 ? 0.4 : 0.1;
        // (the stretch of twos behind the ones saves the gene side's logarithm only: ~0.18 of an entry)
        auto cost = [&](int64_t s) {
            const int32_t f1 = L.slice_fast[s] & 0xFFFF, f2 = L.slice_fast[s] >> 16;
            return (double)L.slice_width[s] - fast_discount * (double)f1 - (side == 0 ? 0.18 : 0.0) * (double)(f2 - f1) + c0;
        };
        std::vector<double> bcost(nblk, 0.0);
        double total = 0.0;
        for (int32_t blk = 0; blk < nblk; blk++) {
            for (int64_t s = bslice0[blk]; s < bslice0[blk + 1]; s++) bcost[blk] += cost(s);
            total += bcost[blk];
        }
        // shares[w] = list of (block, slices) segments of workgroup w
        std::vector<std::vector<std::pair<int32_t, std::vector<int32_t>>>> shares(L.n_wg);
        std::vector<int32_t> live;                            // blocks that have slices
        for (int32_t blk = 0; blk < nblk; blk++) if (bslice0[blk + 1] > bslice0[blk]) live.push_back(blk);
        if ((int64_t)live.size() <= L.n_wg && !live.empty()) {
            std::vector<int> G(nblk, 0);
            std::vector<std::pair<double, int32_t>> frac;
            int used = 0;
            for (int32_t blk : live) {
                double quota = L.n_wg * bcost[blk] / total;
                int g = std::max(1, (int)std::floor(quota));
                g = (int)std::min<int64_t>(g, bslice0[blk + 1] - bslice0[blk]);
                G[blk] = g; used += g;
                frac.emplace_back(quota - g, blk);
            }
            std::stable_sort(frac.begin(), frac.end(), [](const std::pair<double, int32_t> &x, const std::pair<double, int32_t> &y) { return x.first > y.first; });
            for (size_t q = 0; used < L.n_wg && !frac.empty(); q = (q + 1) % frac.size()) {   // hand out the spare workgroups
                int32_t blk = frac[q].second;
                if (G[blk] < bslice0[blk + 1] - bslice0[blk]) { G[blk]++; used++; }
                else if (q + 1 == frac.size()) { bool any = false; for (auto &f : frac) any |= G[f.second] < bslice0[f.second + 1] - bslice0[f.second]; if (!any) break; }
            }
            while (used > L.n_wg) {                           // too many (each block needs at least one): shrink the most over-served
                int32_t worst = -1;
                for (int32_t blk : live) if (G[blk] > 1 && (worst < 0 || bcost[blk] / G[blk] < bcost[worst] / G[worst])) worst = blk;
                if (worst < 0) break;
                G[worst]--; used--;
            }
            int w = 0;
            for (int32_t blk : live) {
                const int g = G[blk];
                const int64_t s0 = bslice0[blk], s1 = bslice0[blk + 1];
                for (int j = 0; j < g; j++) shares[w + j].emplace_back(blk, std::vector<int32_t>());
                {
                    // longest-processing-time deal: slices by cost, descending (ties by id), each to the workgroup
                    // of the block with the least cost so far (ties to the lowest) -- equal cost AND, because the
                    // costly slices go round first, the same mix of long and short slices in every share
                    std::vector<int32_t> by_cost;
                    for (int64_t i = s0; i < s1; i++) by_cost.push_back((int32_t)i);
                    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int32_t x, int32_t y) { return cost(x) > cost(y); });
                    std::vector<double> load(g, 0.0);
                    for (int32_t i : by_cost) {
                        int j = 0;
                        for (int q = 1; q < g; q++) if (load[q] < load[j]) j = q;
                        shares[w + j].back().second.push_back(i);
                        load[j] += cost(i);
                    }
                }
                w += g;
            }
        } else {
            std::vector<int32_t> ord(live);
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return bcost[x] > bcost[y]; });
            std::vector<double> load(L.n_wg, 0.0);
            for (int32_t blk : ord) {
                int best = 0;
                for (int w = 1; w < L.n_wg; w++) if (load[w] < load[best]) best = w;
                shares[best].emplace_back(blk, std::vector<int32_t>());
                for (int64_t i = bslice0[blk]; i < bslice0[blk + 1]; i++) shares[best].back().second.push_back((int32_t)i);
                load[best] += bcost[blk];
            }
            for (auto &sh : shares)
                std::stable_sort(sh.begin(), sh.end(), [](const std::pair<int32_t, std::vector<int32_t>> &x, const std::pair<int32_t, std::vector<int32_t>> &y) { return x.first < y.first; });
        }
        L.wg_seg0.assign(L.n_wg + 1, 0);
        L.seg_block.clear();
        L.seg_ptr.assign(1, 0);
        std::vector<int32_t> order;                           // order[new slice id] = id before the renumbering below
        order.reserve(L.n_slices);
        // Inside a share the waves take slices DYNAMICALLY (an LDS ticket counter), longest first: the
        // hardware issues the oldest wave of a SIMD first, so equal static shares finish far apart
        // (measured: 50 / 75 / 99 us for the three waves of a SIMD) while greedy longest-first pulling
        // ends all waves within one slice of each other.  Which wave runs a slice does not change any
        // result (per-task partials; per-slice evidence partials summed in list order).
        for (int w = 0; w < L.n_wg; w++) {
            L.wg_seg0[w] = (int32_t)L.seg_block.size();
            for (auto &seg : shares[w]) {
                std::vector<int32_t> &sl = seg.second;
                std::stable_sort(sl.begin(), sl.end(), [&](int32_t x, int32_t y) { return L.slice_width[x] > L.slice_width[y]; });
                L.seg_block.push_back(seg.first);
                order.insert(order.end(), sl.begin(), sl.end());
                L.seg_ptr.push_back((int32_t)order.size());
            }
        }
        L.wg_seg0[L.n_wg] = (int32_t)L.seg_block.size();
        L.n_segs = (int64_t)L.seg_block.size();

        // Renumber the slices in processing order, so that list position == slice id: the kernel then finds
        // a slice's width, offset, majors and partial rows directly from its ticket, with no indirection.
        const std::vector<int32_t> &ord = order;
        std::vector<int32_t> w2(L.n_slices), b2(L.n_slices), f2(L.n_slices), len2((size_t)L.n_slices * kLanes), n12((size_t)L.n_slices * kLanes), n22((size_t)L.n_slices * kLanes);
        std::vector<uint32_t> maj2((size_t)L.n_slices * kLanes);
        std::vector<int64_t> pos2((size_t)L.n_slices * kLanes);
        for (int64_t s = 0; s < L.n_slices; s++) {
            const int64_t o = ord[s];
            w2[s] = L.slice_width[o]; b2[s] = L.slice_block[o]; f2[s] = L.slice_fast[o];
            for (int l = 0; l < kLanes; l++) {
                maj2[(size_t)s * kLanes + l] = L.task_major[(size_t)o * kLanes + l];
                pos2[(size_t)s * kLanes + l] = task_pos[(size_t)o * kLanes + l];
                len2[(size_t)s * kLanes + l] = task_len[(size_t)o * kLanes + l];
                n12[(size_t)s * kLanes + l] = task_n1[(size_t)o * kLanes + l];
                n22[(size_t)s * kLanes + l] = task_n2[(size_t)o * kLanes + l];
            }
        }
        L.slice_width.swap(w2); L.slice_block.swap(b2); L.slice_fast.swap(f2); L.task_major.swap(maj2); task_pos.swap(pos2); task_len.swap(len2);
        task_n1.swap(n12); task_n2.swap(n22);
        int64_t o2 = 0;
        for (int64_t s = 0; s < L.n_slices; s++) { L.slice_off[s] = o2; o2 += (int64_t)L.slice_width[s] * kLanes; }
    }
    lap("shares");
    build_inverse();
    lap("inverse");

    try {
        // not zero-filled (the fill below writes every slot, padding included: first touch by the thread that fills)
        if (sink) { if (int rc = sink->place(L)) return rc; }
        else if (L.wide) { L.wide_idx.resize(L.n_slots); L.wide_val.resize(L.n_slots); }
        else L.packed.resize(L.n_slots);
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory building the tiled layout (%lld slots)", (long long)L.n_slots);
    }

    // fill: slot(t, lane) = off + (t/4)*256 + lane*4 + t%4 ; padding slots are {minor 0, value 0}.
    //
    // The order of a task's entries is free (it only fixes the summation order), so it is chosen
    // to keep the LDS gathers of the sweep conflict-free: a ds_read_b128 wave instruction is served
    // in four fixed groups of 16 lanes, one LDS cycle per group when the 16 addresses fall in 16
    // different 16-byte bank slots.  Rows of the staged factor are an odd number of slots long, so
    // the slot of piece p of row `local` is (stride*local + p) mod 16: two lanes of a group collide
    // exactly when their minors are congruent mod 16.  Step by step, the 16 lanes of a group choose in
    // turn (the turn order rotates with the step): a lane takes, among the residues (local mod 16) it
    // still has entries of and no earlier lane of this step took, the one it has most of; a lane that
    // finds all its residues taken doubles up on the least used one.  Round 2 chose residue by residue
    // (demand order, each to the lane with the fewest other residues left): twice the inner work plus
    // a sort per step -- 85 % of the layout's build time -- for conflict rates this scheme undercuts
    // (LDS cycles per group read on the headline matrix, gene / cell side: 1.70 / 1.86 then, 1.66 / 1.83 now).
    lap("alloc");
    static const int kGroupOf[64] = {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,
                                     2,2,2,2,3,3,3,3,3,3,3,3,2,2,2,2,3,3,3,3,2,2,2,2,2,2,2,2,3,3,3,3};
    const bool schedule = env_int("VBNMF_NO_BANK_SCHEDULE", 0) == 0;
    const bool rides = env_int("VBNMF_NO_BANK_RIDES", 0) == 0;            // (A/B switch of round 5's broadcast rides, below)
    // Slices differ in cost by two orders of magnitude and lie sorted by width inside a segment: small chunks
    // handed out through a shared counter, not one contiguous range per thread.
    std::atomic<int64_t> next_chunk{0};
    std::atomic<bool> fill_oom{false};                     // a worker thread that runs out of memory says so here (an exception
    const int64_t kChunk = 16;                             // leaving a std::thread would end the process)
    parallel_for(host_threads(), [&](int64_t, int64_t, int) {
        try {
        // phase 0 = the task's entries of value 1 (placed first, see above), 1 = those of value 2, 2 = the others
        // (every entry when the fast stretch is off)
        constexpr int NP = 3;
        std::vector<int32_t> sorted[16];                   // per lane of the group: entry positions by (phase, residue, minor)
        std::vector<uint8_t> bucket_of;                    // (phase, residue) of every entry of the task at hand
        for (;;) {
        const int64_t c0 = next_chunk.fetch_add(kChunk);
        if (c0 >= L.n_slices) break;
        for (int64_t s = c0; s < std::min<int64_t>(L.n_slices, c0 + kChunk); s++) {
            const int32_t m0 = (int32_t)bstart[L.slice_block[s]];
            const int64_t so = L.slice_off[s];
            const int32_t width = L.slice_width[s];
            auto put = [&](int lane, int64_t t, int64_t q) {
                int64_t slot = so + (t / kUnroll) * (kLanes * kUnroll) + lane * kUnroll + (t % kUnroll);
                uint32_t local = (uint32_t)(idx[q] - m0);
                if (L.wide) { L.wide_idx[slot] = local; L.wide_val[slot] = val[q]; }
                else L.packed[slot] = ((uint32_t)val[q] << kPackedCountShift) | ((local * (uint32_t)L.row_slots) << 4);
            };
            auto pad = [&](int lane, int64_t t0) {         // the arrays are not zero-filled at allocation: the tail of every lane is
                for (int64_t t = t0; t < width; t++) {
                    int64_t slot = so + (t / kUnroll) * (kLanes * kUnroll) + lane * kUnroll + (t % kUnroll);
                    if (L.wide) { L.wide_idx[slot] = 0u; L.wide_val[slot] = 0.0; } else L.packed[slot] = 0u;
                }
            };
            auto phase_of = [&](int64_t q) { return !fast_ones ? 2 : (val[q] == 1.0 ? 0 : (val[q] == 2.0 ? 1 : 2)); };
            if (!schedule) {
                for (int lane = 0; lane < kLanes; lane++) {
                    size_t id = (size_t)s * kLanes + lane;
                    int64_t t = 0;
                    if (L.task_major[id] != kIdleLane)
                        for (int ph = 0; ph < NP; ph++)
                            for (int64_t u = 0; u < task_len[id]; u++)
                                if (phase_of(task_pos[id] + u) == ph) put(lane, t++, task_pos[id] + u);
                    pad(lane, t);
                }
                continue;
            }
            for (int g = 0; g < 4; g++) {
                int lanes[16], nl = 0;
                for (int lane = 0; lane < kLanes; lane++) if (kGroupOf[lane] == g) lanes[nl++] = lane;
                // key[lane][phase][residue] = (entries left << 4) | (15 - residue): the largest key among a lane's candidates is
                // "most entries left, ties to the lowest residue" in one comparison
                uint32_t key[16][NP][16];
                int32_t nxt[16][NP][16];                   // where the next entry of that bucket sits in sorted[lane]
                int32_t nrow[16][NP][16];                  // ... and the local minor (row of the staged block) of that entry
                uint16_t avail[16][NP] = {};               // residues with entries left, as a bit mask
                int32_t rem[16][NP] = {}, step[16] = {};
                int64_t base[16];
                int T = 0;
                for (int j = 0; j < 16; j++) {
                    const size_t id = (size_t)s * kLanes + lanes[j];
                    base[j] = 0;
                    if (L.task_major[id] == kIdleLane) continue;
                    const int64_t q0 = task_pos[id];
                    const int32_t len = task_len[id];
                    base[j] = q0;
                    bucket_of.resize(len);
                    int32_t cnt[NP * 16] = {};
                    for (int32_t t = 0; t < len; t++) {    // one pass over the task: phase and residue of every entry
                        const int b = phase_of(q0 + t) * 16 + ((idx[q0 + t] - m0) & 15);
                        bucket_of[t] = (uint8_t)b;
                        cnt[b]++;
                    }
                    int32_t o = 0, w[NP * 16];
                    for (int ph = 0; ph < NP; ph++)
                        for (int r = 0; r < 16; r++) {
                            const int32_t c = cnt[ph * 16 + r];
                            nxt[j][ph][r] = o; w[ph * 16 + r] = o; o += c;
                            rem[j][ph] += c;
                            key[j][ph][r] = ((uint32_t)c << 4) | (uint32_t)(15 - r);
                            if (c) avail[j][ph] |= (uint16_t)(1u << r);
                        }
                    sorted[j].resize(len);
                    for (int32_t t = 0; t < len; t++)      // stable: a bucket keeps its entries in ascending minor order
                        sorted[j][w[bucket_of[t]]++] = t;
                    for (int ph = 0; ph < NP; ph++)
                        for (int r = 0; r < 16; r++)
                            nrow[j][ph][r] = (key[j][ph][r] >> 4) ? (int32_t)(idx[q0 + sorted[j][nxt[j][ph][r]]] - m0) : -1;
                    T = std::max(T, len);
                }
                for (int t = 0; t < T; t++) {
                    uint32_t used = 0;                     // residues taken in this step
                    uint8_t usedcnt[16] = {};
                    int32_t row_of[16];                    // the row the FIRST taker of a residue reads in this step
                    for (int q = 0; q < 16; q++) {
                        const int j = (t + q) & 15;
                        const int ph = rem[j][0] > 0 ? 0 : (rem[j][1] > 0 ? 1 : 2);
                        if (rem[j][ph] == 0) continue;
                        const uint32_t *k = key[j][ph];
                        // A FREE RIDE first (round 5): lanes of a group that read the SAME row in a step share one address -- a
                        // broadcast, not a conflict.  If the next entry of one of this lane's buckets is the very row an earlier
                        // lane of the step reads, it goes now.  Neighbouring tasks share many minors -- the layout keeps similar
                        // cells together, and a gene's cells recur from gene to gene --: LDS cycles per group read on the headline
                        // matrix 1.33 -> 1.18 (gene side) and 1.76 -> 1.30 (cell side) by the CPU model that reproduces the counters
                        // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.345 before).
                        int ride = -1;
                        for (uint32_t a = rides ? (avail[j][ph] & used) : 0u; a; a &= a - 1) {
                            const int r = __builtin_ctz(a);
                            if (nrow[j][ph][r] == row_of[r]) { ride = r; break; }
                        }
                        uint32_t cand = avail[j][ph] & ~used;
                        int best;
                        if (ride >= 0) best = ride;
                        else if (cand) {
                            uint32_t bk = 0;
                            best = 0;
                            while (cand) {
                                const int r = __builtin_ctz(cand);
                                cand &= cand - 1;
                                if (k[r] > bk) { bk = k[r]; best = r; }
                            }
                        } else {                           // every residue it has is taken: the least used, then the fullest, then the lowest
                            best = -1;
                            for (uint32_t a = avail[j][ph]; a; a &= a - 1) {
                                const int r = __builtin_ctz(a);
                                if (best < 0 || usedcnt[r] < usedcnt[best] || (usedcnt[r] == usedcnt[best] && k[r] > k[best])) best = r;
                            }
                        }
                        put(lanes[j], step[j]++, base[j] + sorted[j][nxt[j][ph][best]++]);
                        if (ride < 0) {
                            if (!((used >> best) & 1u)) row_of[best] = nrow[j][ph][best];
                            used |= 1u << best; usedcnt[best]++;
                        }
                        rem[j][ph]--;
                        key[j][ph][best] -= 16;
                        if ((key[j][ph][best] >> 4) == 0) { avail[j][ph] &= (uint16_t)~(1u << best); nrow[j][ph][best] = -1; }
                        else nrow[j][ph][best] = (int32_t)(idx[base[j] + sorted[j][nxt[j][ph][best]]] - m0);
                    }
                }
                for (int j = 0; j < 16; j++) pad(lanes[j], step[j]);
            }
        }
        }
        } catch (const std::bad_alloc &) {
            fill_oom.store(true);
            next_chunk.store(L.n_slices);                  // the other workers stop at their next chunk
        }
    });
    if (fill_oom.load()) return fail(VBNMF_ERR_OOM, "out of host memory filling the tiled layout");
    lap("fill");
    return VBNMF_OK;
}

std::shared_ptr<const Layout> shared_layout(const vbnmf_matrix *X, int side, const LayoutParams &lp, int &rc, LayoutSink *sink, bool *built)
{
    rc = VBNMF_OK;
    if (built) *built = false;
    const int cap = 2 * std::max(0, env_int("VBNMF_LAYOUT_CACHE", 3));       // entries = pairs x 2 sides
    LayoutCache &C = X->layouts;
    auto same = [&](const LayoutCache::Entry &q) {
        return q.side == side && q.lp.block_width == lp.block_width && q.lp.block_cap == lp.block_cap && q.lp.max_len == lp.max_len && q.lp.n_wg == lp.n_wg &&
               q.lp.row_slots == lp.row_slots;
    };
    {
        std::lock_guard<std::mutex> g(C.mu);
        for (size_t i = 0; i < C.entries.size(); i++)
            if (same(C.entries[i])) {
                LayoutCache::Entry hit = C.entries[i];
                C.entries.erase(C.entries.begin() + i);
                C.entries.push_back(hit);                                  // most recently used last
                return hit.layout;
            }
    }
    if (X->M.shell) {
        rc = fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries and no imported layout of side %d "
                  "for this geometry (block %d, row stride %d slots, %d workgroups); import it with vbnmf_matrix_import_layout", side,
                  lp.block_width, lp.row_slots, lp.n_wg);
        return nullptr;
    }
    auto L = std::make_shared<Layout>();
    rc = build_layout(X->M, 0, X->M.m, side, lp, &X->M.cell_order(), *L, sink);
    if (rc) return nullptr;
    if (built) *built = true;
    if (cap > 0 || sink) {
        std::lock_guard<std::mutex> g(C.mu);
        C.entries.push_back({side, lp, L});
        while ((int)C.entries.size() > std::max(cap, 2)) {
            const Layout *gone = C.entries.front().layout.get();
            C.entries.erase(C.entries.begin());
            for (size_t i = 0; i < C.copies.size();)          // its device copies live on only in the engines that use them
                if (C.copies[i].key == gone) C.copies.erase(C.copies.begin() + i); else i++;
        }
    }
    return L;
}

void cache_layout(const vbnmf_matrix *X, int side, const LayoutParams &lp, std::shared_ptr<const Layout> L)
{
    // an imported layout is kept whatever VBNMF_LAYOUT_CACHE says (a shell cannot rebuild it); it counts towards the cap
    const int cap = std::max(2, 2 * std::max(0, env_int("VBNMF_LAYOUT_CACHE", 3)));
    LayoutCache &C = X->layouts;
    std::lock_guard<std::mutex> g(C.mu);
    for (size_t i = 0; i < C.entries.size();) {
        const auto &q = C.entries[i];
        const bool same = q.side == side && q.lp.block_width == lp.block_width && q.lp.block_cap == lp.block_cap && q.lp.max_len == lp.max_len &&
                          q.lp.n_wg == lp.n_wg && q.lp.row_slots == lp.row_slots;
        if (same) C.entries.erase(C.entries.begin() + i); else i++;
    }
    C.entries.push_back({side, lp, std::move(L)});
    while (!X->M.shell && (int)C.entries.size() > cap) {          // (a shell cannot cut an evicted layout again: it keeps them all)
        const Layout *gone = C.entries.front().layout.get();
        C.entries.erase(C.entries.begin());
        for (size_t i = 0; i < C.copies.size();)
            if (C.copies[i].key == gone) C.copies.erase(C.copies.begin() + i); else i++;
    }
}

std::vector<int32_t> rank_classes(const int32_t *ranks, int32_t count, int32_t max_classes)
{
    std::vector<int32_t> padded;
    for (int32_t q = 0; q < count; q++) padded.push_back(padded_rank(ranks[q]));
    std::sort(padded.begin(), padded.end());
    padded.erase(std::unique(padded.begin(), padded.end()), padded.end());
    std::vector<int32_t> classes;
    if (!padded.empty()) {
        if (max_classes < 1) max_classes = 1;
        int32_t top = padded.back();
        classes.push_back(top);
        while ((int32_t)classes.size() < max_classes) {
            // the largest planned rank whose rows are at most half as wide as the current lowest class's
            int32_t next = 0;
            for (int32_t p : padded) if (lds_row_bytes(p) * 2 <= lds_row_bytes(top)) next = p;
            if (!next) break;
            classes.push_back(next);
            top = next;
        }
        std::sort(classes.begin(), classes.end());
    }
    return classes;
}

int plan_class(const vbnmf_matrix *X, int R)
{
    std::lock_guard<std::mutex> g(X->plan_mu);
    for (int32_t c : X->plan) if (c >= R) return c;
    return R;
}

std::shared_ptr<void> cached_device_copy(const vbnmf_matrix *X, const Layout *L, int device)
{
    LayoutCache &C = X->layouts;
    std::lock_guard<std::mutex> g(C.mu);
    for (const auto &c : C.copies)
        if (c.key == L && c.device == device) return c.arrays;
    return nullptr;
}

void store_device_copy(const vbnmf_matrix *X, const Layout *L, int device, std::shared_ptr<void> arrays)
{
    LayoutCache &C = X->layouts;
    std::lock_guard<std::mutex> g(C.mu);
    bool cached = false;
    for (const auto &q : C.entries) cached |= q.layout.get() == L;
    if (!cached) return;
    for (const auto &c : C.copies)
        if (c.key == L && c.device == device) return;         // another thread was first
    C.copies.push_back({L, device, std::move(arrays)});
}

}  // namespace vbnmf

// ====================================================================== C ABI (host-only part)
using namespace vbnmf;

extern "C" {

const char *vbnmf_last_error(void) { return vbnmf::last_error_cstr(); }
const char *vbnmf_version(void) { return "0.1.0"; }

}  // extern "C"

int vbnmf::new_matrix(vbnmf_matrix **out, const std::function<int(Matrix &)> &fill)
{
    if (!out) return fail(VBNMF_ERR_BAD_ARG, "out pointer is NULL");
    *out = nullptr;
    vbnmf_matrix *X = nullptr;
    try {
        X = new vbnmf_matrix();
        int rc = fill(X->M);
        if (rc) { delete X; return rc; }
        X->lgx = sum_lgamma_x1(X->M, 0, X->M.m);
    } catch (const std::bad_alloc &) {
        delete X;
        return fail(VBNMF_ERR_OOM, "out of host memory ingesting X");
    } catch (const std::exception &ex) {
        delete X;
        return fail(VBNMF_ERR_BAD_ARG, "ingesting X failed: %s", ex.what());
    }
    *out = X;
    return VBNMF_OK;
}

extern "C" {

static int check_dims(int64_t n, int64_t m)
{
    if (n <= 0 || m <= 0) return fail(VBNMF_ERR_BAD_ARG, "matrix dimensions must be positive (got %lld x %lld)", (long long)n, (long long)m);
    if (n > 0x7FFFFFFFLL - 64 || m > 0x7FFFFFFFLL - 64) return fail(VBNMF_ERR_BAD_ARG, "a matrix dimension exceeds 2^31-65");
    return VBNMF_OK;
}

int vbnmf_matrix_from_dense(int64_t n, int64_t m, const double *A, vbnmf_matrix **out)
{
    if (int rc = check_dims(n, m)) return rc;
    if (!A) return fail(VBNMF_ERR_BAD_ARG, "X is NULL");
    return new_matrix(out, [&](Matrix &M) { return matrix_from_dense(n, m, A, M); });
}

int vbnmf_matrix_from_csc(int64_t n, int64_t m, const int32_t *p, const int32_t *i, const double *x, vbnmf_matrix **out)
{
    if (int rc = check_dims(n, m)) return rc;
    if (!p || ((!i || !x) && p[m] > 0)) return fail(VBNMF_ERR_BAD_ARG, "a CSC slot pointer is NULL");
    return new_matrix(out, [&](Matrix &M) { return matrix_from_csc(n, m, p, i, x, M); });
}

int vbnmf_matrix_from_csr(int64_t n, int64_t m, const int32_t *p, const int32_t *j, const double *x, vbnmf_matrix **out)
{
    if (int rc = check_dims(n, m)) return rc;
    if (!p || ((!j || !x) && p[n] > 0)) return fail(VBNMF_ERR_BAD_ARG, "a CSR slot pointer is NULL");
    return new_matrix(out, [&](Matrix &M) { return matrix_from_csr(n, m, p, j, x, M); });
}

int vbnmf_matrix_info(const vbnmf_matrix *X, int64_t *n, int64_t *m, int64_t *nnz, double *lgx)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (n) *n = X->M.n;
    if (m) *m = X->M.m;
    if (nnz) *nnz = X->M.nnz;
    if (lgx) *lgx = X->lgx;
    return VBNMF_OK;
}

int vbnmf_matrix_empty_counts(const vbnmf_matrix *X, int64_t *empty_rows, int64_t *empty_cols)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (X->M.shell) return fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries");
    const Matrix &M = X->M;
    // rowSums(mat)==0 / colSums(mat)==0 of reference R/bayesian.R:244-245 (sums, not stored-entry counts).  Columns are
    // cut in chunks with their own row-sum arrays, added in chunk order (a fixed order: the test for == 0 must not
    // depend on the thread count); 0.15 s single-threaded at the headline size, once per vb_factorize call.
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(64, M.nnz / (1 << 20) + 1));       // chunks: a function of X alone
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, ((int64_t)1 << 24) / std::max<int64_t>(1, M.n)));   // <= 128 MB of row sums in all
    std::vector<std::vector<double>> part(T);
    std::vector<int64_t> ecs(T, 0);
    try {                                                   // allocated HERE, not in the worker threads: a bad_alloc there would end in std::terminate
        for (auto &rs : part) rs.assign(M.n, 0.0);
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory checking for empty rows (%d x %lld doubles)", T, (long long)M.n);
    }
    parallel_for(T, [&](int64_t b, int64_t e, int) {
        for (int64_t c = b; c < e; c++) {
            std::vector<double> &rs = part[c];
            for (int64_t j = M.m * c / T; j < M.m * (c + 1) / T; j++) {
                double cs = 0.0;
                for (int64_t q = M.colptr[j]; q < M.colptr[j + 1]; q++) { cs += M.val[q]; rs[M.row[q]] += M.val[q]; }
                ecs[c] += (cs == 0.0);
            }
        }
    });
    int64_t ec = 0, er = 0;
    for (int c = 0; c < T; c++) ec += ecs[c];
    for (int64_t i = 0; i < M.n; i++) {
        double v = 0.0;
        for (int c = 0; c < T; c++) v += part[c][i];
        er += (v == 0.0);
    }
    if (empty_rows) *empty_rows = er;
    if (empty_cols) *empty_cols = ec;
    return VBNMF_OK;
}

// Rank classes of a sweep over several ranks (reference R/bayesian.R:316: `for(rank in ranks)`, every rank on the same
// matrix).  The tiled layout depends on the rank only through the LDS row size; cutting one per row size costs more host
// time than the whole sweep spends on the device (BASELINE config C4: six geometries, 5 s, against 0.3 s of stepping).
// With a plan, every rank uses the geometry of the smallest class at or above it: narrower blocks than its own rows
// would allow (more, shorter tasks: a slower step), but cut once.  max_classes = 1: one class at the largest rank;
// k > 1: the k - 1 further classes halve the remaining range of row sizes each (largest first).  count = 0 clears it.
int vbnmf_matrix_plan_ranks(vbnmf_matrix *X, const int32_t *ranks, int32_t count, int32_t max_classes)
{
    if (!X || (count > 0 && !ranks) || count < 0) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    for (int32_t q = 0; q < count; q++)
        if (ranks[q] < 1 || ranks[q] > VBNMF_MAX_RANK) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [1, %d]", ranks[q], VBNMF_MAX_RANK);
    std::vector<int32_t> classes = rank_classes(ranks, count, max_classes);
    std::lock_guard<std::mutex> g(X->plan_mu);
    X->plan.swap(classes);
    return VBNMF_OK;
}

void vbnmf_matrix_destroy(vbnmf_matrix *X) { delete X; }

int vbnmf_layout_build(const vbnmf_matrix *X, int64_t col_begin, int64_t col_end, int32_t side, int32_t r,
                       vbnmf_layout **out, vbnmf_layout_view *view)
{
    if (!X || !out || !view) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (side != 0 && side != 1) return fail(VBNMF_ERR_BAD_ARG, "side must be 0 or 1");
    if (r < 1 || r > VBNMF_MAX_RANK) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [1, %d]", r, VBNMF_MAX_RANK);
    if (X->M.shell) return fail(VBNMF_ERR_STATE, "this matrix handle is a shell (vbnmf_matrix_shell): it holds no entries");
    *out = nullptr;
    vbnmf_layout *H = nullptr;
    try {
        H = new vbnmf_layout();
        int R = std::max(padded_rank(r), plan_class(X, padded_rank(r)));      // the geometry of the rank's class (plan_ranks)
        int64_t nmaj = side == 0 ? X->M.n : col_end - col_begin;
        int64_t nmin = side == 0 ? col_end - col_begin : X->M.n;
        const bool range_ok = col_begin >= 0 && col_end <= X->M.m && col_begin < col_end;      // build_layout reports a bad range
        LayoutParams lp = default_layout_params(nmaj, nmin, R, 0, range_ok ? X->M.colptr[col_end] - X->M.colptr[col_begin] : 0);
        // the cells in the order an engine on these columns uses (order.cpp): the matrix's for the whole matrix, the range's own otherwise
        std::vector<int32_t> own;
        const bool whole = col_begin == 0 && col_end == X->M.m;
        if (!whole && range_ok) own = compute_cell_order(X->M, col_begin, col_end);
        int rc = build_layout(X->M, col_begin, col_end, side, lp, whole ? &X->M.cell_order() : &own, H->L);
        if (rc) { delete H; return rc; }
    } catch (const std::bad_alloc &) {
        delete H;
        return fail(VBNMF_ERR_OOM, "out of host memory building the layout");
    }
    const Layout &L = H->L;
    view->side = L.side; view->wide = L.wide ? 1 : 0;
    view->n_major = L.n_major; view->n_minor = L.n_minor;
    view->block_width = L.block_width; view->n_blocks = L.n_blocks; view->max_len = L.max_len; view->n_wg = L.n_wg;
    view->n_tasks = L.n_tasks; view->n_slices = L.n_slices; view->n_slots = L.n_slots; view->n_segs = L.n_segs;
    view->task_major = L.task_major.data(); view->slice_width = L.slice_width.data();
    view->slice_off = L.slice_off.data(); view->slice_block = L.slice_block.data(); view->slice_fast = L.slice_fast.data();
    view->seg_block = L.seg_block.data(); view->wg_seg0 = L.wg_seg0.data();
    view->seg_ptr = L.seg_ptr.data(); view->row_slots = L.row_slots; view->block_start = L.block_start.data();
    view->inv_ptr = L.inv_ptr.data(); view->inv_task = L.inv_task.data();
    view->packed = L.wide ? nullptr : L.packed.data();
    view->wide_idx = L.wide ? L.wide_idx.data() : nullptr;
    view->wide_val = L.wide ? L.wide_val.data() : nullptr;
    view->cell_perm = L.cell_perm.empty() ? nullptr : L.cell_perm.data();
    *out = H;
    return VBNMF_OK;
}

void vbnmf_layout_destroy(vbnmf_layout *L) { delete L; }

}  // extern "C"

// ---------------------------------------------------------------- node-shared layouts (one build per node, not per process)
//
// The reference ships the whole `bundle` -- the matrix included -- to every MPI slave (reference R/bayesian.R:252-263) and
// every slave densifies it again per iteration.  Here the processes of one node (one per GPU) share ONE ingestion and
// ONE pair of tiled layouts: the process that holds X exports a layout as a flat blob (into shared memory the caller
// maps), the others import it into a matrix SHELL -- a handle with X's metadata and no entries -- and upload it to their
// own GPU.  Blob = header (int64 words) + the layout's arrays, each 64-byte aligned, + a closing magic word.
namespace {

constexpr int64_t kBlobMagic = 0x56424E4D464C5930LL;      // "VBNMFLY0"
constexpr int64_t kBlobVersion = 2;
constexpr int kBlobHeaderWords = 48;
constexpr int kBlobArrays = 15;

struct BlobArray { const void *src; void *dst; int64_t bytes; };

inline int64_t align64(int64_t v) { return (v + 63) & ~(int64_t)63; }

template <class V> int64_t vec_bytes(const V &v) { return (int64_t)v.size() * (int64_t)sizeof(typename V::value_type); }

// the arrays of a layout in blob order (pointers valid while L is)
void blob_arrays(const Layout &L, BlobArray (&a)[kBlobArrays])
{
    int q = 0;
    auto put = [&](const auto &v) { a[q].src = v.data(); a[q].dst = nullptr; a[q].bytes = vec_bytes(v); q++; };
    put(L.task_major); put(L.slice_width); put(L.slice_off); put(L.slice_block); put(L.slice_fast); put(L.block_start);
    put(L.seg_block); put(L.wg_seg0); put(L.seg_ptr); put(L.inv_ptr); put(L.inv_task); put(L.packed); put(L.wide_idx); put(L.wide_val);
    put(L.cell_perm);
}

void parallel_copy(void *dst, const void *src, int64_t bytes)
{
    const int64_t chunk = (int64_t)4 << 20;
    const int64_t nchunks = (bytes + chunk - 1) / chunk;
    parallel_for(nchunks, [&](int64_t b, int64_t e, int) {
        for (int64_t c = b; c < e; c++) {
            const int64_t o = c * chunk, len = std::min(chunk, bytes - o);
            std::memcpy(static_cast<char *>(dst) + o, static_cast<const char *>(src) + o, (size_t)len);
        }
    });
}

// geometry of the whole-matrix layout of `side` that an engine of rank `geometry_rank` with n_wg workgroups uses
LayoutParams whole_matrix_params(const vbnmf_matrix *X, int side, int geometry_rank, int n_wg)
{
    const int R = padded_rank(geometry_rank);
    const int64_t nmaj = side == 0 ? X->M.n : X->M.m, nmin = side == 0 ? X->M.m : X->M.n;
    return default_layout_params(nmaj, nmin, R, n_wg, X->M.nnz);
}

}  // namespace

extern "C" {

int32_t vbnmf_padded_rank(int32_t r) { return (r < 1 || r > VBNMF_MAX_RANK) ? 0 : padded_rank(r); }
int32_t vbnmf_host_threads(void) { return host_threads(); }
int32_t vbnmf_set_host_threads(int32_t n)
{
    const int32_t before = host_threads();
    set_host_threads_override(n);
    return before;
}

// Rank classes of a sweep (see vbnmf_matrix_plan_ranks) WITHOUT touching a matrix: classes[0..n) = padded ranks, ascending.
int vbnmf_plan_classes(const int32_t *ranks, int32_t count, int32_t max_classes, int32_t *classes, int32_t *n_classes)
{
    if ((count > 0 && !ranks) || count < 0 || !classes || !n_classes) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    for (int32_t q = 0; q < count; q++)
        if (ranks[q] < 1 || ranks[q] > VBNMF_MAX_RANK) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [1, %d]", ranks[q], VBNMF_MAX_RANK);
    const std::vector<int32_t> c = rank_classes(ranks, count, max_classes);
    for (size_t q = 0; q < c.size(); q++) classes[q] = c[q];       // at most `count` entries
    *n_classes = (int32_t)c.size();
    return VBNMF_OK;
}

// meta[8] = n, m, stored entries, counts_int, counts_u16, max value, sum lgamma(x+1), sum(-x log x + x)
int vbnmf_matrix_get_meta(const vbnmf_matrix *X, double *meta)
{
    if (!X || !meta) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (!X->M.shell) std::call_once(X->xlx_once, [&] { X->xlx = sum_xlogx(X->M, 0, X->M.m); });
    meta[0] = (double)X->M.n; meta[1] = (double)X->M.m; meta[2] = (double)X->M.nnz;
    meta[3] = X->M.counts_int ? 1.0 : 0.0; meta[4] = X->M.counts_u16 ? 1.0 : 0.0; meta[5] = X->M.max_val;
    meta[6] = X->lgx; meta[7] = X->xlx;
    return VBNMF_OK;
}

int vbnmf_matrix_shell(const double *meta, vbnmf_matrix **out)
{
    if (!meta || !out) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    const int64_t n = (int64_t)meta[0], m = (int64_t)meta[1], nnz = (int64_t)meta[2];
    if (int rc = check_dims(n, m)) return rc;
    if (nnz < 0) return fail(VBNMF_ERR_BAD_ARG, "negative entry count");
    vbnmf_matrix *X = new (std::nothrow) vbnmf_matrix();
    if (!X) return fail(VBNMF_ERR_OOM, "out of host memory");
    X->M.shell = true;
    X->M.n = n; X->M.m = m; X->M.nnz = nnz;
    X->M.counts_int = meta[3] != 0.0; X->M.counts_u16 = meta[4] != 0.0; X->M.max_val = meta[5];
    X->lgx = meta[6];
    std::call_once(X->xlx_once, [&] { X->xlx = meta[7]; });
    *out = X;
    return VBNMF_OK;
}

int vbnmf_matrix_is_shell(const vbnmf_matrix *X) { return X && X->M.shell ? 1 : 0; }

// The per-matrix work every whole-matrix layout starts from, done ahead of need (e.g. on a second host thread while the
// cell side is being cut): the order of the cells (order.cpp) and the row-major copy the gene side is cut from.
// The same on a background host thread owned by the handle (joined by vbnmf_matrix_prepare and by destroy): the call
// returns at once, and whoever needs the order or the row-major copy first simply waits for it (std::call_once).
int vbnmf_matrix_prepare_async(const vbnmf_matrix *X)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (X->M.shell) return VBNMF_OK;
    std::lock_guard<std::mutex> g(X->prep_mu);
    if (X->prep.joinable()) return VBNMF_OK;                     // already under way (or done, not yet joined)
    try {
        X->prep = std::thread([X] {
            try { (void)X->M.cell_order(); (void)X->M.row_major(); } catch (...) { /* the consumer that needs them reports the failure */ }
        });
    } catch (const std::system_error &) {
        return fail(VBNMF_ERR_OOM, "could not start the background thread");
    }
    return VBNMF_OK;
}

int vbnmf_matrix_prepare(const vbnmf_matrix *X)
{
    if (!X) return fail(VBNMF_ERR_BAD_ARG, "matrix handle is NULL");
    if (X->M.shell) return VBNMF_OK;
    {
        std::lock_guard<std::mutex> g(X->prep_mu);
        if (X->prep.joinable()) X->prep.join();
    }
    try {
        (void)X->M.cell_order();
        (void)X->M.row_major();
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory preparing the matrix");
    }
    return VBNMF_OK;
}

// Blob of the whole-matrix layout of `side` in the geometry of rank `geometry_rank` for engines with n_wg sweep
// workgroups (vbnmf_device_sweep_workgroups).  buf == NULL: builds (and caches) the layout and returns its blob size in
// *bytes; otherwise writes the blob (all host threads) into buf[0..capacity).
int vbnmf_matrix_export_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg, void *buf,
                               int64_t capacity, int64_t *bytes)
{
    if (!X || !bytes) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (side != 0 && side != 1) return fail(VBNMF_ERR_BAD_ARG, "side must be 0 or 1");
    if (geometry_rank < 1 || geometry_rank > VBNMF_MAX_RANK) return fail(VBNMF_ERR_BAD_ARG, "rank %d is outside [1, %d]", geometry_rank, VBNMF_MAX_RANK);
    if (n_wg < 1) return fail(VBNMF_ERR_BAD_ARG, "n_wg must be positive");
    std::shared_ptr<const Layout> L;
    LayoutParams lp;
    try {
        lp = whole_matrix_params(X, side, geometry_rank, n_wg);
        int rc = VBNMF_OK;
        L = shared_layout(X, side, lp, rc);
        if (rc) return rc;
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory building the layout");
    }
    BlobArray a[kBlobArrays];
    blob_arrays(*L, a);
    int64_t total = kBlobHeaderWords * 8;
    for (int q = 0; q < kBlobArrays; q++) total = align64(total) + a[q].bytes;
    total = align64(total) + 8;
    *bytes = total;
    if (!buf) return VBNMF_OK;
    if (capacity < total) return fail(VBNMF_ERR_BAD_ARG, "the buffer holds %lld bytes, the layout blob needs %lld", (long long)capacity, (long long)total);
    int64_t *h = static_cast<int64_t *>(buf);
    std::memset(h, 0, kBlobHeaderWords * 8);
    h[0] = kBlobMagic; h[1] = kBlobVersion; h[2] = total;
    h[3] = L->side; h[4] = L->wide ? 1 : 0; h[5] = L->n_major; h[6] = L->n_minor; h[7] = L->block_width; h[8] = L->n_blocks;
    h[9] = L->max_len; h[10] = L->n_wg; h[11] = L->row_slots; h[12] = L->n_tasks; h[13] = L->n_slices; h[14] = L->n_slots;
    h[15] = L->n_segs; h[16] = L->nnz;
    h[17] = lp.block_width; h[18] = lp.block_cap; h[19] = lp.max_len; h[20] = lp.n_wg; h[21] = lp.row_slots;
    h[22] = X->M.n; h[23] = X->M.m; h[24] = X->M.nnz;
    for (int q = 0; q < kBlobArrays; q++) h[32 + q] = a[q].bytes;
    int64_t off = kBlobHeaderWords * 8;
    for (int q = 0; q < kBlobArrays; q++) {
        off = align64(off);
        parallel_copy(static_cast<char *>(buf) + off, a[q].src, a[q].bytes);
        off += a[q].bytes;
    }
    off = align64(off);
    std::memcpy(static_cast<char *>(buf) + off, &kBlobMagic, 8);
    return VBNMF_OK;
}

// Adds the layout in buf[0..bytes) (written by vbnmf_matrix_export_layout, this library version) to X's cache: engines
// created afterwards in that geometry use it instead of cutting their own.  X: a shell or a full handle of the same matrix.
}  // extern "C"

namespace {
// keep == null: every array is copied out of the blob; otherwise the big arrays (entry stream) stay where they are --
// inside a mapping that `keep` holds for as long as the layout lives.
int load_blob(const vbnmf_matrix *X, const void *buf, int64_t bytes, std::shared_ptr<void> keep)
{
    if (!X || !buf) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (bytes < kBlobHeaderWords * 8 + 8) return fail(VBNMF_ERR_BAD_ARG, "layout blob is truncated (%lld bytes)", (long long)bytes);
    const int64_t *h = static_cast<const int64_t *>(buf);
    if (h[0] != kBlobMagic || h[1] != kBlobVersion) return fail(VBNMF_ERR_BAD_ARG, "not a layout blob of this library version");
    if (h[2] != bytes) return fail(VBNMF_ERR_BAD_ARG, "layout blob says %lld bytes, %lld were handed in", (long long)h[2], (long long)bytes);
    if (h[22] != X->M.n || h[23] != X->M.m || h[24] != X->M.nnz)
        return fail(VBNMF_ERR_BAD_ARG, "the layout blob was cut from a %lld x %lld matrix with %lld entries, this handle is %lld x %lld with %lld",
                    (long long)h[22], (long long)h[23], (long long)h[24], (long long)X->M.n, (long long)X->M.m, (long long)X->M.nnz);
    if (h[3] != 0 && h[3] != 1) return fail(VBNMF_ERR_BAD_ARG, "layout blob: bad side");
    int64_t off = kBlobHeaderWords * 8;
    for (int q = 0; q < kBlobArrays; q++) {
        if (h[32 + q] < 0) return fail(VBNMF_ERR_BAD_ARG, "layout blob: negative array size");
        off = align64(off) + h[32 + q];
        if (off > bytes) return fail(VBNMF_ERR_BAD_ARG, "layout blob: arrays run past the end");
    }
    off = align64(off);
    int64_t tail = 0;
    if (off + 8 != bytes) return fail(VBNMF_ERR_BAD_ARG, "layout blob: size does not match its array table");
    std::memcpy(&tail, static_cast<const char *>(buf) + off, 8);
    if (tail != kBlobMagic) return fail(VBNMF_ERR_BAD_ARG, "layout blob: closing word missing (a partial write?)");
    try {
        auto L = std::make_shared<Layout>();
        L->side = (int)h[3]; L->wide = h[4] != 0; L->n_major = h[5]; L->n_minor = h[6]; L->block_width = (int32_t)h[7];
        L->n_blocks = (int32_t)h[8]; L->max_len = (int32_t)h[9]; L->n_wg = (int32_t)h[10]; L->row_slots = (int32_t)h[11];
        L->n_tasks = h[12]; L->n_slices = h[13]; L->n_slots = h[14]; L->n_segs = h[15]; L->nnz = h[16];
        LayoutParams lp;
        lp.block_width = (int32_t)h[17]; lp.block_cap = (int32_t)h[18]; lp.max_len = (int32_t)h[19]; lp.n_wg = (int32_t)h[20]; lp.row_slots = (int32_t)h[21];
        int q = 0;
        int64_t o = kBlobHeaderWords * 8;
        int rc = VBNMF_OK;
        auto take = [&](auto &v) {
            using T = typename std::remove_reference<decltype(v)>::type::value_type;
            o = align64(o);
            const int64_t nb = h[32 + q];
            if (nb % (int64_t)sizeof(T)) rc = fail(VBNMF_ERR_BAD_ARG, "layout blob: array %d has a ragged size", q);
            else {
                v.resize((size_t)(nb / (int64_t)sizeof(T)));
                parallel_copy(v.data(), static_cast<const char *>(buf) + o, nb);
            }
            o += nb; q++;
        };
        auto take_big = [&](auto &v) {                       // copied, or adopted in place when the blob is a kept mapping
            using T = typename std::remove_reference<decltype(v)>::type::value_type;
            if (!keep) { take(v); return; }
            o = align64(o);
            const int64_t nb = h[32 + q];
            if (nb % (int64_t)sizeof(T)) rc = fail(VBNMF_ERR_BAD_ARG, "layout blob: array %d has a ragged size", q);
            else v.adopt(reinterpret_cast<T *>(const_cast<char *>(static_cast<const char *>(buf) + o)), (size_t)(nb / (int64_t)sizeof(T)), keep);
            o += nb; q++;
        };
        take(L->task_major); take(L->slice_width); take(L->slice_off); take(L->slice_block); take(L->slice_fast); take(L->block_start);
        take(L->seg_block); take(L->wg_seg0); take(L->seg_ptr); take(L->inv_ptr); take(L->inv_task);
        take_big(L->packed); take_big(L->wide_idx); take_big(L->wide_val);
        take(L->cell_perm);
        if (rc) return rc;
        // the renumbering of the cells: a permutation, and the SAME one for every layout of this matrix (the engine's
        // cell-indexed arrays live in it); a shell adopts the first one it sees
        if (!L->cell_perm.empty()) {
            if ((int64_t)L->cell_perm.size() != X->M.m) return fail(VBNMF_ERR_BAD_ARG, "layout blob: cell order of the wrong length");
            std::vector<char> seen(X->M.m, 0);
            for (int32_t v : L->cell_perm) {
                if (v < 0 || v >= X->M.m || seen[v]) return fail(VBNMF_ERR_BAD_ARG, "layout blob: the cell order is not a permutation");
                seen[v] = 1;
            }
        }
        std::call_once(X->M.order_cache->once, [&] { X->M.order_cache->perm = L->cell_perm; });
        if (X->M.order_cache->perm != L->cell_perm)
            return fail(VBNMF_ERR_BAD_ARG, "layout blob: its cell order differs from the one this matrix handle already uses");
        // the scalar fields must agree with the arrays they describe (the kernels index by them)
        const bool ok = (int64_t)L->task_major.size() == L->n_slices * kLanes && (int64_t)L->slice_width.size() == L->n_slices &&
                        (int64_t)L->slice_off.size() == L->n_slices && (int64_t)L->slice_fast.size() == L->n_slices &&
                        (int64_t)L->block_start.size() == (int64_t)L->n_blocks + 1 && (int64_t)L->seg_block.size() == L->n_segs &&
                        (int64_t)L->wg_seg0.size() == (int64_t)L->n_wg + 1 && (int64_t)L->seg_ptr.size() == L->n_segs + 1 &&
                        (int64_t)L->inv_ptr.size() == L->n_major + 1 && (int64_t)L->inv_task.size() == L->n_tasks &&
                        (L->wide ? ((int64_t)L->wide_idx.size() == L->n_slots && (int64_t)L->wide_val.size() == L->n_slots)
                                 : (int64_t)L->packed.size() == L->n_slots) &&
                        L->n_major == (L->side == 0 ? X->M.n : X->M.m) && L->n_minor == (L->side == 0 ? X->M.m : X->M.n) &&
                        L->wide == !X->M.counts_int;
        if (!ok) return fail(VBNMF_ERR_BAD_ARG, "layout blob: header and arrays disagree");
        // ... and the CONTENTS of the small arrays are what the kernels index device memory by: a blob from another build with
        // the same version word, or a half-overwritten mapping, must be an error here, not an out-of-bounds access on the GPU.
        // (The entry stream itself addresses LDS rows only: its offsets are masked to the staged block.)
        {
            const char *bad = nullptr;
            const int64_t nsl = L->n_slices, nseg = L->n_segs;
            if (L->n_blocks < 1 || L->n_wg < 1 || L->row_slots < 1 || !(L->row_slots & 1) || L->max_len < 4 || L->block_width < 1) bad = "geometry";
            for (int64_t q = 0; !bad && q < nsl; q++) {
                const int64_t w = L->slice_width[q], o = L->slice_off[q];
                if (w < 4 || (w & 3) || w > L->max_len + 3 || o < 0 || (o & 255) || o + w * kLanes > L->n_slots) bad = "slice_off / slice_width";
                else if ((L->slice_fast[q] & 0xFFFF) > w || ((L->slice_fast[q] >> 16) & 0xFFFF) > w) bad = "slice_fast";
            }
            for (size_t q = 0; !bad && q < L->task_major.size(); q++)
                if (L->task_major[q] != kIdleLane && (int64_t)L->task_major[q] >= L->n_major) bad = "task_major";
            if (!bad && (L->block_start[0] != 0 || L->block_start[L->n_blocks] != L->n_minor)) bad = "block_start";
            for (int q = 0; !bad && q < L->n_blocks; q++) {
                const int64_t w = L->block_start[q + 1] - L->block_start[q];
                if (w < 1 || w > L->block_width) bad = "block_start";
            }
            for (int64_t q = 0; !bad && q < nseg; q++) if (L->seg_block[q] < 0 || L->seg_block[q] >= L->n_blocks) bad = "seg_block";
            if (!bad && (L->seg_ptr[0] != 0 || L->seg_ptr[nseg] != nsl)) bad = "seg_ptr";
            for (int64_t q = 0; !bad && q < nseg; q++) if (L->seg_ptr[q + 1] < L->seg_ptr[q]) bad = "seg_ptr";
            if (!bad && (L->wg_seg0[0] != 0 || L->wg_seg0[L->n_wg] != nseg)) bad = "wg_seg0";
            for (int q = 0; !bad && q < L->n_wg; q++) if (L->wg_seg0[q + 1] < L->wg_seg0[q]) bad = "wg_seg0";
            if (!bad && (L->inv_ptr[0] != 0 || L->inv_ptr[L->n_major] != L->n_tasks)) bad = "inv_ptr";
            for (int64_t q = 0; !bad && q < L->n_major; q++) if (L->inv_ptr[q + 1] < L->inv_ptr[q]) bad = "inv_ptr";
            for (int64_t q = 0; !bad && q < L->n_tasks; q++) if ((int64_t)L->inv_task[q] >= nsl * kLanes) bad = "inv_task";
            if (bad) return fail(VBNMF_ERR_BAD_ARG, "layout blob: the %s array is inconsistent (another build, or a damaged file?)", bad);
        }
        cache_layout(X, L->side, lp, L);
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory importing the layout");
    }
    return VBNMF_OK;
}

struct ShmMap {
    void *base = nullptr;
    size_t bytes = 0;
    ~ShmMap() { if (base) munmap(base, bytes); }
};

// The sink of vbnmf_matrix_share_layout: at the point where build_layout knows every size, create `path`.part with the
// whole blob's size and hand the layout its big arrays INSIDE the mapping.
struct ShmSink : LayoutSink {
    std::string path;
    std::shared_ptr<ShmMap> map;
    int64_t total = 0;
    int64_t offs[kBlobArrays] = {};
    int place(Layout &L) override
    {
        // sizes of every array in blob order (the big ones from n_slots: they are not allocated yet)
        BlobArray a[kBlobArrays];
        blob_arrays(L, a);
        const int iP = 11, iWI = 12, iWV = 13;               // packed, wide_idx, wide_val in blob order
        a[iP].bytes = L.wide ? 0 : L.n_slots * (int64_t)sizeof(uint32_t);
        a[iWI].bytes = L.wide ? L.n_slots * (int64_t)sizeof(uint32_t) : 0;
        a[iWV].bytes = L.wide ? L.n_slots * (int64_t)sizeof(double) : 0;
        int64_t off = kBlobHeaderWords * 8;
        for (int q = 0; q < kBlobArrays; q++) { off = align64(off); offs[q] = off; off += a[q].bytes; }
        total = align64(off) + 8;
        {   // a memory file system that is full answers the WRITES with SIGBUS, not the ftruncate with an error: ask first
            std::string dir = path.substr(0, path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/'));
            if (dir.empty()) dir = ".";
            struct statvfs vs;
            if (statvfs(dir.c_str(), &vs) == 0 && (double)vs.f_bavail * (double)vs.f_frsize < (double)total)
                return fail(VBNMF_ERR_OOM, "%s has %.0f MB free, the layout needs %.0f MB", dir.c_str(),
                            (double)vs.f_bavail * (double)vs.f_frsize / 1e6, (double)total / 1e6);
        }
        const std::string part = path + ".part";
        const int fd = open(part.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(VBNMF_ERR_BAD_ARG, "cannot create %s: %s", part.c_str(), strerror(errno));
        if (ftruncate(fd, (off_t)total) != 0) { close(fd); unlink(part.c_str()); return fail(VBNMF_ERR_OOM, "cannot size %s to %lld bytes: %s", part.c_str(), (long long)total, strerror(errno)); }
        void *base = mmap(nullptr, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base == MAP_FAILED) { unlink(part.c_str()); return fail(VBNMF_ERR_OOM, "cannot map %s: %s", part.c_str(), strerror(errno)); }
        map = std::make_shared<ShmMap>();
        map->base = base; map->bytes = (size_t)total;
        char *b = static_cast<char *>(base);
        if (L.wide) {
            L.wide_idx.adopt(reinterpret_cast<uint32_t *>(b + offs[iWI]), (size_t)L.n_slots, map);
            L.wide_val.adopt(reinterpret_cast<double *>(b + offs[iWV]), (size_t)L.n_slots, map);
        } else {
            L.packed.adopt(reinterpret_cast<uint32_t *>(b + offs[iP]), (size_t)L.n_slots, map);
        }
        return VBNMF_OK;
    }
};

void write_blob_header(int64_t *h, const Layout &L, const LayoutParams &lp, const vbnmf_matrix *X, int64_t total, const BlobArray (&a)[kBlobArrays])
{
    std::memset(h, 0, kBlobHeaderWords * 8);
    h[0] = kBlobMagic; h[1] = kBlobVersion; h[2] = total;
    h[3] = L.side; h[4] = L.wide ? 1 : 0; h[5] = L.n_major; h[6] = L.n_minor; h[7] = L.block_width; h[8] = L.n_blocks;
    h[9] = L.max_len; h[10] = L.n_wg; h[11] = L.row_slots; h[12] = L.n_tasks; h[13] = L.n_slices; h[14] = L.n_slots;
    h[15] = L.n_segs; h[16] = L.nnz;
    h[17] = lp.block_width; h[18] = lp.block_cap; h[19] = lp.max_len; h[20] = lp.n_wg; h[21] = lp.row_slots;
    h[22] = X->M.n; h[23] = X->M.m; h[24] = X->M.nnz;
    for (int q = 0; q < kBlobArrays; q++) h[32 + q] = a[q].bytes;
}

}  // namespace

extern "C" {

int vbnmf_matrix_import_layout(const vbnmf_matrix *X, const void *buf, int64_t bytes)
{
    return load_blob(X, buf, bytes, nullptr);
}

// The same layout, but living ONCE in the node's shared memory.  share: cuts the layout with its big arrays written
// straight into a new file `path` (a tmpfs path, e.g. under /dev/shm; built as path + ".part" and renamed when complete,
// so a peer that sees `path` sees all of it) and keeps that mapping as the layout's storage; a layout that is already
// cached in ordinary memory is copied into the file instead.  attach: maps `path` read-only and adopts the big arrays in
// place (the small index arrays are copied).  The file may be unlinked as soon as every process has attached.
int vbnmf_matrix_share_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg, const char *path)
{
    if (!X || !path) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    if (side != 0 && side != 1) return fail(VBNMF_ERR_BAD_ARG, "side must be 0 or 1");
    if (geometry_rank < 1 || geometry_rank > VBNMF_MAX_RANK || n_wg < 1) return fail(VBNMF_ERR_BAD_ARG, "bad geometry");
    try {
        const LayoutParams lp = whole_matrix_params(X, side, geometry_rank, n_wg);
        ShmSink sink;
        sink.path = path;
        int rc = VBNMF_OK;
        bool built = false;
        std::shared_ptr<const Layout> L = shared_layout(X, side, lp, rc, &sink, &built);
        if (rc) { if (sink.map) unlink((sink.path + ".part").c_str()); return rc; }
        BlobArray a[kBlobArrays];
        blob_arrays(*L, a);
        if (!built || !sink.map) {
            // already cached in ordinary memory: write a copy (the copying export into a fresh file)
            int64_t total = kBlobHeaderWords * 8;
            for (int q = 0; q < kBlobArrays; q++) total = align64(total) + a[q].bytes;
            total = align64(total) + 8;
            const std::string part = std::string(path) + ".part";
            const int fd = open(part.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return fail(VBNMF_ERR_BAD_ARG, "cannot create %s: %s", part.c_str(), strerror(errno));
            if (ftruncate(fd, (off_t)total) != 0) { close(fd); unlink(part.c_str()); return fail(VBNMF_ERR_OOM, "cannot size %s: %s", part.c_str(), strerror(errno)); }
            void *base = mmap(nullptr, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (base == MAP_FAILED) { unlink(part.c_str()); return fail(VBNMF_ERR_OOM, "cannot map %s: %s", part.c_str(), strerror(errno)); }
            int64_t nb = 0;
            rc = vbnmf_matrix_export_layout(X, side, geometry_rank, n_wg, base, total, &nb);
            munmap(base, (size_t)total);
            if (rc) { unlink(part.c_str()); return rc; }
            if (rename(part.c_str(), path) != 0) { unlink(part.c_str()); return fail(VBNMF_ERR_BAD_ARG, "cannot rename %s: %s", part.c_str(), strerror(errno)); }
            return VBNMF_OK;
        }
        // header, the small arrays and the closing word around the big arrays that build_layout already wrote in place
        char *b = static_cast<char *>(sink.map->base);
        write_blob_header(reinterpret_cast<int64_t *>(b), *L, lp, X, sink.total, a);
        for (int q = 0; q < kBlobArrays; q++) {
            if (a[q].src == b + sink.offs[q]) continue;                      // a big array: in place already
            if (a[q].bytes) std::memcpy(b + sink.offs[q], a[q].src, (size_t)a[q].bytes);
        }
        std::memcpy(b + sink.total - 8, &kBlobMagic, 8);
        const std::string part = sink.path + ".part";
        if (rename(part.c_str(), path) != 0) { unlink(part.c_str()); return fail(VBNMF_ERR_BAD_ARG, "cannot rename %s: %s", part.c_str(), strerror(errno)); }
    } catch (const std::bad_alloc &) {
        return fail(VBNMF_ERR_OOM, "out of host memory building the layout");
    }
    return VBNMF_OK;
}

int vbnmf_matrix_attach_layout(const vbnmf_matrix *X, const char *path)
{
    if (!X || !path) return fail(VBNMF_ERR_BAD_ARG, "NULL argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(VBNMF_ERR_BAD_ARG, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < kBlobHeaderWords * 8 + 8) { close(fd); return fail(VBNMF_ERR_BAD_ARG, "%s is not a layout blob", path); }
    void *base = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return fail(VBNMF_ERR_OOM, "cannot map %s: %s", path, strerror(errno));
    auto map = std::make_shared<ShmMap>();
    map->base = base; map->bytes = (size_t)st.st_size;
    return load_blob(X, base, (int64_t)st.st_size, map);
}

}  // extern "C"
