// order.cpp -- a label-free renumbering of the cells, internal to the tiled layout (host side, no device code).
//
// Why.  A task of the sweep is one major's entries inside ONE minor block (a block = the factor rows that fit the
// workgroup's 160 KB of LDS: 1 996 cells at rank 10, 906 at rank 20), and every task costs a factor-row load, a
// log-factor-row load, a partial-statistics row written by k_sweep and gathered again by k_update -- 480 B at rank 20
// against ~170 B of entries for the average task there.  On the gene side the number of tasks is the number of non-empty
// (gene, cell-block) pairs.  Single-cell counts have cluster structure: a gene is expressed in some cell types and
// (nearly) silent in others -- the reference's own simulator draws one gene distribution per cluster
// (reference R/utils.R:787-795).  With the cells of a type stored next to each other a gene's entries fall into FEWER
// blocks: measured on the headline matrix (profiles/r04_order_ab.txt) the gene side goes from 535 k to 383 k tasks at
// rank 10 and from 1.21 M to 0.58 M at rank 20, the whole step from 512 to 452 us at rank 20.
//
// The step is permutation-equivariant, so the renumbering is invisible outside the library: the device holds the
// cell-indexed arrays in the new order, and the boundary (set_state / get_state / ids / SVD vectors / random_state's
// per-element counters) translates through `perm`.
//
// How (no labels are available, and none are needed -- any grouping that puts cells with alike gene support next to
// each other helps, and splitting a true cluster in two costs nothing as long as the halves stay adjacent):
//   1. sketch: every cell's sqrt-counts summed over D = 64 pseudo-random gene groups, normalised to unit length
//      (one pass over the stored entries);
//   2. spherical k-means on the sketches, K <= 32 centroids, farthest-point start on a fixed sample, 6 rounds;
//   3. the clusters are chained by centroid similarity (greedy nearest neighbour, then 2-opt), so that neighbours in
//      the order share gene support and a block that straddles two clusters still sees alike cells;
//   4. cells are sorted by their cluster's place in the chain, original order inside a cluster (stable).
// Everything is deterministic and independent of the host thread count (fixed chunking, fixed summation order): every
// process that orders the same columns gets the same permutation.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace vbnmf {

namespace {

constexpr int kD = 64;            // gene groups of the sketch
constexpr int kMaxK = 32;         // centroids
constexpr int kRounds = 6;
constexpr int64_t kChunk = 1024;  // cells per reduction chunk (fixed: the sums do not depend on the thread count)

inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// eight partial sums in a fixed order: the compiler may keep them in one vector register without re-associating anything
inline float dot(const float *a, const float *b)
{
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < kD; q += 8)
        for (int u = 0; u < 8; u++) s[u] += a[q + u] * b[q + u];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

}  // namespace

// 0: off, 1: forced on (any size), -1: automatic (matrices where the gene side has several blocks at every rank)
static int order_mode()
{
    const char *s = getenv("VBNMF_CELL_ORDER");
    if (!s || !*s) return -1;
    return atoi(s) != 0 ? 1 : 0;
}

std::vector<int32_t> compute_cell_order(const Matrix &X, int64_t cb, int64_t ce)
{
    const int64_t m = ce - cb, n = X.n;
    const int mode = order_mode();
    // automatic: only where a gene's entries can spread over several cell blocks (the widest block holds ~20 000 rows at
    // rank 1), the clusters have enough members for the sketches to say anything, and the matrix is sparse enough for a
    // (gene, block) pair to be EMPTY sometimes (at 75 % density every pair is taken whatever the order)
    const int64_t stored = X.colptr[ce] - X.colptr[cb];
    if (mode == 0 || m < 64 || (mode < 0 && (m < 8192 || (double)stored > 0.25 * (double)n * (double)m))) return {};
    int K = (int)std::min<int64_t>(kMaxK, std::max<int64_t>(2, m / 256));
    int rounds = kRounds;
    if (const char *sv = getenv("VBNMF_ORDER_K")) { const int v = atoi(sv); if (v >= 2 && v <= 256) K = (int)std::min<int64_t>(v, std::max<int64_t>(2, m / 8)); }   // experiments
    if (const char *sv = getenv("VBNMF_ORDER_ROUNDS")) { const int v = atoi(sv); if (v >= 1 && v <= 100) rounds = v; }

    // 1. sketches
    std::vector<uint8_t> group(n);
    for (int64_t i = 0; i < n; i++) group[i] = (uint8_t)(mix64((uint64_t)i) % kD);
    std::vector<float> S((size_t)m * kD);
    parallel_for(m, [&](int64_t b, int64_t e, int) {
        for (int64_t j = b; j < e; j++) {
            float *s = &S[(size_t)j * kD];
            for (int q = 0; q < kD; q++) s[q] = 0.f;
            for (int64_t q = X.colptr[cb + j]; q < X.colptr[cb + j + 1]; q++) s[group[X.row[q]]] += std::sqrt((float)std::fabs(X.val[q]));
            float nn = 0.f;
            for (int q = 0; q < kD; q++) nn += s[q] * s[q];
            if (nn > 0.f) { const float inv = 1.f / std::sqrt(nn); for (int q = 0; q < kD; q++) s[q] *= inv; }
        }
    });

    // 2. k-means.  Start: farthest-point on a fixed pseudo-random sample (the next centroid is the sample cell least
    // alike every centroid chosen so far), so that a start does not hold two centroids of one cluster while another
    // cluster has none.
    const int64_t ns = std::min<int64_t>(m, 4096);
    std::vector<int64_t> sample(ns);
    for (int64_t q = 0; q < ns; q++) sample[q] = ns == m ? q : (int64_t)(mix64(0xC0FFEEull + (uint64_t)q) % (uint64_t)m);
    std::vector<float> C((size_t)K * kD);
    std::vector<float> best(ns, -2.f);                       // similarity to the closest chosen centroid
    int64_t pick = sample[0];
    for (int k = 0; k < K; k++) {
        std::memcpy(&C[(size_t)k * kD], &S[(size_t)pick * kD], kD * sizeof(float));
        float low = 3.f;
        int64_t arg = 0;
        for (int64_t q = 0; q < ns; q++) {
            best[q] = std::max(best[q], dot(&S[(size_t)sample[q] * kD], &C[(size_t)k * kD]));
            if (best[q] < low) { low = best[q]; arg = q; }
        }
        pick = sample[arg];
    }
    std::vector<int32_t> label(m, 0);
    const int64_t nchunks = (m + kChunk - 1) / kChunk;
    std::vector<float> part((size_t)nchunks * K * kD);
    std::vector<int32_t> cnt((size_t)nchunks * K);
    auto assign = [&](bool accumulate) {
        parallel_for(nchunks, [&](int64_t c0, int64_t c1, int) {
            for (int64_t c = c0; c < c1; c++) {
                float *pc = &part[(size_t)c * K * kD];
                int32_t *nc = &cnt[(size_t)c * K];
                if (accumulate) { std::fill(pc, pc + (size_t)K * kD, 0.f); std::fill(nc, nc + K, 0); }
                for (int64_t j = c * kChunk; j < std::min(m, (c + 1) * kChunk); j++) {
                    const float *s = &S[(size_t)j * kD];
                    int bk = 0;
                    float bv = -2.f;
                    for (int k = 0; k < K; k++) { const float v = dot(s, &C[(size_t)k * kD]); if (v > bv) { bv = v; bk = k; } }
                    label[j] = bk;
                    if (accumulate) { float *t = pc + (size_t)bk * kD; for (int q = 0; q < kD; q++) t[q] += s[q]; nc[bk]++; }
                }
            }
        });
    };
    for (int round = 0; round < rounds; round++) {
        assign(true);
        for (int k = 0; k < K; k++) {
            float acc[kD] = {};
            int64_t members = 0;
            for (int64_t c = 0; c < nchunks; c++) {           // chunk order: fixed
                const float *t = &part[((size_t)c * K + k) * kD];
                for (int q = 0; q < kD; q++) acc[q] += t[q];
                members += cnt[(size_t)c * K + k];
            }
            float nn = 0.f;
            for (int q = 0; q < kD; q++) nn += acc[q] * acc[q];
            if (members > 0 && nn > 0.f) {
                const float inv = 1.f / std::sqrt(nn);
                for (int q = 0; q < kD; q++) C[(size_t)k * kD + q] = acc[q] * inv;
            }                                                 // an empty cluster keeps its centroid
        }
    }
    assign(false);

    // 3. chain the clusters: greedy nearest neighbour from the largest cluster, then 2-opt on the path length
    std::vector<int64_t> size(K, 0);
    for (int64_t j = 0; j < m; j++) size[label[j]]++;
    std::vector<float> sim((size_t)K * K);
    for (int a = 0; a < K; a++) for (int b = 0; b < K; b++) sim[(size_t)a * K + b] = dot(&C[(size_t)a * kD], &C[(size_t)b * kD]);
    std::vector<int> chain;
    std::vector<char> used(K, 0);
    int cur = (int)(std::max_element(size.begin(), size.end()) - size.begin());
    chain.push_back(cur); used[cur] = 1;
    for (int step = 1; step < K; step++) {
        int nxt = -1;
        for (int k = 0; k < K; k++) if (!used[k] && (nxt < 0 || sim[(size_t)cur * K + k] > sim[(size_t)cur * K + nxt])) nxt = k;
        chain.push_back(nxt); used[nxt] = 1; cur = nxt;
    }
    auto link = [&](int a, int b) { return 1.f - sim[(size_t)chain[a] * K + chain[b]]; };
    for (int pass = 0; pass < 8; pass++) {
        bool improved = false;
        for (int a = 0; a + 1 < K; a++)
            for (int b = a + 1; b < K; b++) {
                // reversing chain[a+1 .. b] replaces links (a, a+1) and (b, b+1) by (a, b) and (a+1, b+1)
                const float before = link(a, a + 1) + (b + 1 < K ? link(b, b + 1) : 0.f);
                const float after = link(a, b) + (b + 1 < K ? link(a + 1, b + 1) : 0.f);
                if (after + 1e-6f < before) { std::reverse(chain.begin() + a + 1, chain.begin() + b + 1); improved = true; }
            }
        if (!improved) break;
    }
    std::vector<int> place(K);
    for (int q = 0; q < K; q++) place[chain[q]] = q;

    // 4. stable counting sort of the cells by their cluster's place
    std::vector<int64_t> start(K + 1, 0);
    for (int k = 0; k < K; k++) start[place[k] + 1] = size[k];
    for (int q = 0; q < K; q++) start[q + 1] += start[q];
    std::vector<int32_t> perm(m);
    for (int64_t j = 0; j < m; j++) perm[start[place[label[j]]]++] = (int32_t)j;
    bool identity = true;
    for (int64_t j = 0; j < m && identity; j++) identity = perm[j] == j;
    if (identity) perm.clear();
    return perm;
}

const std::vector<int32_t> &Matrix::cell_order() const
{
    std::call_once(order_cache->once, [&] { if (!shell) order_cache->perm = compute_cell_order(*this, 0, m); });
    return order_cache->perm;
}

}  // namespace vbnmf
