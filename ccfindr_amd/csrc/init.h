// init.h -- device kernels either side of the update path (included once, by engine.hip):
//   * the 'random' initial state of vb_init (reference R/bayesian.R:111-115: w ~ Gamma(shape aw, scale bw/aw),
//     h ~ Gamma(ah, bh/ah)) drawn on the device: Philox4x32-10 counters + Marsaglia-Tsang.  R's RNG stream cannot be
//     matched outside R; the draws are a function of (seed, factor, element) only, so they do not depend on the
//     launch geometry, the device or the rank count of a partitioned run;
//   * the consensus stopping count of factorize() (reference R/factorize.R:198-208: sum(cnn != cnn0) over all pairs
//     of cells, cnn from connectivity(), :51-60) without the O(m^2) vectors: a contingency table of the old and new
//     arg-max labels, pairs(row sums) + pairs(column sums) - 2 pairs(table);
//   * the dense pieces of the truncated SVD behind the 'svd2' initialiser (reference R/bayesian.R:150-159, irlba):
//     Gram matrix of a tall [N][R] block, Cholesky factor / Jacobi eigen-decomposition of the k x k result, and the
//     product of the tall block with a k x k matrix -- so that the subspace stays on the device between the sparse
//     products (k_spmm) and nothing crosses PCIe inside the iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vbnmf {

// ---------------------------------------------------------------- Philox4x32-10
struct Philox {
    uint32_t c[4], k[2];
};
__host__ __device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t (&k)[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// 4 x 32 random bits for counter (i0, i1, i2, i3) under key (k0, k1)
__host__ __device__ __forceinline__ void philox4x32(uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t k0, uint32_t k1,
                                                    uint32_t (&out)[4])
{
    uint32_t c[4] = {i0, i1, i2, i3}, k[2] = {k0, k1};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// uniform in (0, 1): 53 bits, never 0 or 1
__host__ __device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{
    const uint64_t v = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);              // 53 bits
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// One Gamma(shape a, scale s) variate for stream (element, factor) of key seed.  Marsaglia & Tsang (2000): for a >= 1,
// d = a - 1/3, c = 1/sqrt(9d): x ~ N(0,1), v = (1 + c x)^3, accept if v > 0 and log u < x^2/2 + d - d v + d log v;
// a < 1: Gamma(a + 1) u^(1/a).  Attempt t of the stream uses counter (element lo, element hi, factor, t).
__device__ inline double gamma_draw(double a, double scale, uint64_t element, uint32_t factor, uint32_t k0, uint32_t k1)
{
    const double a1 = a < 1.0 ? a + 1.0 : a;
    const double d = a1 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = 0.0;
    uint32_t r[4];
    for (uint32_t t = 0; t < 64; t++) {                       // acceptance > 0.95 per attempt: 64 never runs out in practice
        philox4x32((uint32_t)element, (uint32_t)(element >> 32), factor, 2 * t, k0, k1, r);
        const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);     // Box-Muller
        philox4x32((uint32_t)element, (uint32_t)(element >> 32), factor, 2 * t + 1, k0, k1, r);
        const double u = u53(r[0], r[1]);
        const double v1 = 1.0 + c * x;
        if (v1 <= 0.0) continue;
        const double v = v1 * v1 * v1;
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) {
            g = d * v;
            if (a < 1.0) g *= pow(u53(r[2], r[3]), 1.0 / a);
            break;
        }
    }
    return g * scale;
}

// f[major][k] = e[major][k] = Gamma(shape a, scale b / a) for k < r (pad columns 0); factor 0 = W, 1 = H.  The element
// index is the GLOBAL (major, k) position (major0 = first cell of a partition), so a partitioned run draws the same H.
__global__ __launch_bounds__(256) void k_gamma_init(double *__restrict__ f, double *__restrict__ e, double *__restrict__ dvar,
                                                    int64_t nmaj, int64_t major0, int r, int R, double a, double b,
                                                    uint32_t factor, uint32_t k0, uint32_t k1, const int32_t *__restrict__ perm)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nmaj * R) return;
    const int64_t M = i / R;
    const int k = (int)(i - M * R);
    double v = 0.0;
    // (perm: the layout's order of the cells -- row M of the array is the caller's major perm[M]; the draw is keyed by the
    // CALLER's global element index, so it does not depend on the order, the launch geometry or the partitioning)
    if (k < r) v = gamma_draw(a, b / a, (uint64_t)(major0 + (perm ? (int64_t)perm[M] : M)) * (uint64_t)r + (uint64_t)k, factor, k0, k1);
    f[i] = v;
    if (e) e[i] = v;
    if (dvar) dvar[i] = 0.0;
}

// standard normal block for the range finder of the truncated SVD: g[major][k], k < r
__global__ __launch_bounds__(256) void k_normal_init(double *__restrict__ g, int64_t nmaj, int r, int R, uint32_t k0, uint32_t k1)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nmaj * R) return;
    const int64_t M = i / R;
    const int k = (int)(i - M * R);
    double v = 0.0;
    if (k < r) {
        uint32_t q[4];
        const uint64_t el = (uint64_t)M * (uint64_t)r + (uint64_t)k;
        philox4x32((uint32_t)el, (uint32_t)(el >> 32), 7u, 0u, k0, k1, q);
        v = sqrt(-2.0 * log(u53(q[0], q[1]))) * cos(6.283185307179586476925 * u53(q[2], q[3]));
    }
    g[i] = v;
}

// ---------------------------------------------------------------- connectivity change count
// table[old][new] += 1 per cell (integer atomics: order-independent).  ids are 1-based, 0 = no label (all-NaN column).
__global__ __launch_bounds__(256) void k_label_table(const int32_t *__restrict__ ids_old, const int32_t *__restrict__ ids_new,
                                                     int64_t m, int r, unsigned long long *__restrict__ table)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int a = ids_old[j], b = ids_new[j];
    atomicAdd(&table[(size_t)a * (r + 1) + b], 1ull);
}
// pairs together in exactly one of the two labelings = pairs(rows) + pairs(cols) - 2 pairs(cells of the table)
__global__ void k_label_pairs(const unsigned long long *__restrict__ table, int r, unsigned long long *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int q = r + 1;
    auto pairs = [](unsigned long long c) { return c * (c - (c > 0 ? 1ull : 0ull)) / 2ull; };
    unsigned long long both = 0, rows = 0, cols = 0;
    for (int a = 0; a < q; a++) {
        unsigned long long rs = 0;
        for (int b = 0; b < q; b++) { const unsigned long long c = table[(size_t)a * q + b]; rs += c; both += pairs(c); }
        rows += pairs(rs);
    }
    for (int b = 0; b < q; b++) {
        unsigned long long cs = 0;
        for (int a = 0; a < q; a++) cs += table[(size_t)a * q + b];
        cols += pairs(cs);
    }
    out[0] = rows + cols - 2ull * both;
}

// ---------------------------------------------------------------- dense pieces of the truncated SVD
constexpr int kGramBlocks = 256;
// Block partials of G = t(A) A for a tall A[N][R]: gp[block][R*R].  Thread (i, j) of a 32 x 32 block owns the elements
// (i + 32 a, j + 32 b) of G, a, b < ceil(R / 32): one element up to R = 32, four at the padded ranks 40..64.
template <int R>
__global__ __launch_bounds__(1024) void k_gram(const double *__restrict__ A, int64_t N, double *__restrict__ gp)
{
    constexpr int T = (R + 31) / 32;
    __shared__ double tile[32][R];
    const int i = threadIdx.x / 32, j = threadIdx.x % 32;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t n0 = (int64_t)blockIdx.x * per, n1 = min(N, n0 + per);
    double acc[T][T];
#pragma unroll
    for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b < T; b++) acc[a][b] = 0.0;
    for (int64_t b0 = n0; b0 < n1; b0 += 32) {
        const int rows = (int)min((int64_t)32, n1 - b0);
        __syncthreads();
        for (int t = threadIdx.x; t < rows * R; t += 1024) tile[t / R][t % R] = A[(size_t)b0 * R + t];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < T; a++)
#pragma unroll
            for (int b = 0; b < T; b++) {
                const int ii = i + 32 * a, jj = j + 32 * b;
                if (ii < R && jj < R)
                    for (int q = 0; q < rows; q++) acc[a][b] = fma(tile[q][ii], tile[q][jj], acc[a][b]);
            }
    }
#pragma unroll
    for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b < T; b++) {
            const int ii = i + 32 * a, jj = j + 32 * b;
            if (ii < R && jj < R) gp[(size_t)blockIdx.x * R * R + ii * R + jj] = acc[a][b];
        }
}

// One block.  G = sum of the block partials (fixed order), then, by `mode`:
//   0: Cholesky G = t(U) U (upper), S = inverse of U (so that A S is orthonormal); diag(U) goes to `vals`
//   1: Jacobi eigen-decomposition G = V diag(lambda) t(V), eigenvalues DESCENDING in `vals`, S = V and, when S2 is
//      given, S2 = V diag(1 / sqrt(lambda)) (right singular vectors: rows of P V / sigma)
// k = columns in use (<= R); columns k..R-1 of S are 0.  status[0] is SET to 1 when G is not positive definite (mode 0).
// vals_host (optional): a pinned, device-visible copy of vals (+ vals_host[R] = seq, written last) for the host's
// convergence test -- no memcpy.
template <int R>
__global__ __launch_bounds__(1024) void k_small(const double *__restrict__ gp, int nb, int k, int mode, double *__restrict__ S,
                                                double *__restrict__ S2, double *__restrict__ vals, double *__restrict__ vals_host,
                                                double seq, int32_t *__restrict__ status)
{
    __shared__ double G[R][R + 1], V[R][R + 1];
    const int i = threadIdx.x / 32, j = threadIdx.x % 32;
    // thread (i, j) of the 32 x 32 block owns the elements (i + 32 a, j + 32 b) of the R x R matrices (R up to 64)
    for (int ii = i; ii < R; ii += 32)
        for (int jj = j; jj < R; jj += 32) {
            double s = 0.0;
            for (int b = 0; b < nb; b++) s += gp[(size_t)b * R * R + ii * R + jj];
            G[ii][jj] = (ii < k && jj < k) ? s : (ii == jj ? 1.0 : 0.0);
            V[ii][jj] = ii == jj ? 1.0 : 0.0;
        }
    __syncthreads();
    if (mode == 0) {
        // Cholesky (upper) by thread 0 of each column step; k <= 64, sequential dependencies: one thread is enough
        if (threadIdx.x == 0) {
            int bad = 0;
            for (int c = 0; c < k; c++) {
                double dsum = G[c][c];
                for (int q = 0; q < c; q++) dsum -= V[q][c] * V[q][c];
                if (!(dsum > 0.0)) { bad = 1; break; }
                const double dd = sqrt(dsum);
                V[c][c] = dd;
                for (int t = c + 1; t < k; t++) {
                    double s2 = G[c][t];
                    for (int q = 0; q < c; q++) s2 -= V[q][c] * V[q][t];
                    V[c][t] = s2 / dd;
                }
                for (int t = 0; t < c; t++) V[c][t] = 0.0;
            }
            if (bad) status[0] = 1;                           // sticky: the caller clears it before a run
            // inverse of the upper factor into G (back substitution, column by column)
            for (int c = 0; c < R; c++) for (int t = 0; t < R; t++) G[c][t] = 0.0;
            if (!bad)
                for (int c = 0; c < k; c++) {
                    G[c][c] = 1.0 / V[c][c];
                    for (int t = c - 1; t >= 0; t--) {
                        double s2 = 0.0;
                        for (int q = t + 1; q <= c; q++) s2 += V[t][q] * G[q][c];
                        G[t][c] = -s2 / V[t][t];
                    }
                }
            for (int c = 0; c < k; c++) { vals[c] = V[c][c]; if (vals_host) vals_host[c] = V[c][c]; }
        }
        __syncthreads();
        for (int ii = i; ii < R; ii += 32)
            for (int jj = j; jj < R; jj += 32) S[ii * R + jj] = G[ii][jj];
        return;
    }
    // cyclic Jacobi on the k x k symmetric G; rotations applied by the 32 lanes of row 0 of the block (lane j: columns j, j + 32)
    __shared__ double off;
    for (int sweep = 0; sweep < 30; sweep++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            double o = 0.0;
            for (int p = 0; p < k; p++) for (int q = p + 1; q < k; q++) o += G[p][q] * G[p][q];
            off = o;
        }
        __syncthreads();
        double tr = 0.0;
        for (int p = 0; p < k; p++) tr += fabs(G[p][p]);
        if (off <= 1e-30 * tr * tr) break;
        for (int p = 0; p < k - 1; p++)
            for (int q = p + 1; q < k; q++) {
                const double apq = G[p][q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double tau = (G[q][q] - G[p][p]) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t); s = t * c;
                }
                __syncthreads();
                if (i == 0)
                    for (int jj = j; jj < k; jj += 32) {         // columns p, q of G and V: G <- G J, V <- V J
                        const double gp_ = G[jj][p], gq_ = G[jj][q];
                        G[jj][p] = c * gp_ - s * gq_; G[jj][q] = s * gp_ + c * gq_;
                        const double vp = V[jj][p], vq = V[jj][q];
                        V[jj][p] = c * vp - s * vq; V[jj][q] = s * vp + c * vq;
                    }
                __syncthreads();
                if (i == 0)
                    for (int jj = j; jj < k; jj += 32) {         // rows p, q of G: G <- t(J) G
                        const double gp_ = G[p][jj], gq_ = G[q][jj];
                        G[p][jj] = c * gp_ - s * gq_; G[q][jj] = s * gp_ + c * gq_;
                    }
                __syncthreads();
            }
    }
    __syncthreads();
    // order the eigenvalues descending (thread 0: selection sort of k <= 64 columns)
    __shared__ int perm[R];
    if (threadIdx.x == 0) {
        for (int c = 0; c < R; c++) perm[c] = c;
        for (int a = 0; a < k; a++) {
            int best = a;
            for (int b = a + 1; b < k; b++) if (G[perm[b]][perm[b]] > G[perm[best]][perm[best]]) best = b;
            const int t = perm[a]; perm[a] = perm[best]; perm[best] = t;
        }
        for (int c = 0; c < k; c++) { vals[c] = G[perm[c]][perm[c]]; if (vals_host) vals_host[c] = vals[c]; }
        if (vals_host) { __threadfence_system(); vals_host[R] = seq; }       // sequence number: the host sees a complete set
    }
    __syncthreads();
    for (int ii = i; ii < R; ii += 32)
        for (int jj = j; jj < R; jj += 32) {
            const double v = (ii < k && jj < k) ? V[ii][perm[jj]] : 0.0;
            S[ii * R + jj] = v;
            if (S2) { const double lam = jj < k ? G[perm[jj]][perm[jj]] : 0.0; S2[ii * R + jj] = lam > 0.0 ? v / sqrt(lam) : 0.0; }
        }
}

// B[N][R] = A[N][R] S (S: R x R row-major); in place allowed (a thread owns a whole row).
template <int R>
__global__ __launch_bounds__(256) void k_apply(const double *__restrict__ A, const double *__restrict__ S, int64_t N, double *__restrict__ B)
{
    __shared__ double sS[R * R];
    for (int t = threadIdx.x; t < R * R; t += 256) sS[t] = S[t];
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    double a[R], o[R];
#pragma unroll
    for (int q = 0; q < R; q++) { a[q] = A[(size_t)row * R + q]; o[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < R; q++)
#pragma unroll
        for (int c = 0; c < R; c++) o[c] = fma(a[q], sS[q * R + c], o[c]);
#pragma unroll
    for (int c = 0; c < R; c++) B[(size_t)row * R + c] = o[c];
}

}  // namespace vbnmf
