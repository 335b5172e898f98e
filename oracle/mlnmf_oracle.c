/*
 * mlnmf_oracle.c -- CPU restatement of ccfindR's maximum-likelihood NMF step on stored entries.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ccfindr_amd/, the HIP library,
 * include/) may link, load or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference (hjunwoo/ccfindR) holds no tests, golden vectors or stored
 * outputs for this path, and R is not in this image, so the reference's own R code cannot be run.
 * This file is the stored-entries form of the step; the dense literal restatement of the same R
 * lines is oracle/mlnmf_oracle.py (numpy), and tests/test_oracle_mlnmf.py holds the two together.
 *
 * Reference lines restated (R/factorize.R):
 *   :8-15   up = h * (t(w) %*% (x/(w %*% h))) ; down = colSums(w) ; [prior: up + a - 1, down + a/b] ;
 *           h = up/down ; h[h < eps] = eps
 *   :17-24  the same for w with the NEW h ; down = rowSums(h)
 *   :40-49  likelihood = ( sum(x*log(wh) - wh) + sum_{x>0}(-x*log(x) + x) ) / n / m
 * x/(w h) is only needed where x != 0 (0/wh = 0 for the finite positive wh the updates keep), and
 * sum(wh) = sum_k colSum(w)_k rowSum(h)_k; both are used here, so results equal the dense form up to
 * summation order.
 *
 * Column-major doubles as R holds them: w n x r ([i + k*n]), h r x m ([k + j*r]); X as dgCMatrix slots.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ML_EPS 2.220446049250313e-16   /* .Machine$double.eps */

int oracle_mlnmf_update_csc(int64_t n, int64_t m, int32_t r,
                            const int32_t *p, const int32_t *ri, const double *x,
                            const double *w_in, const double *h_in,
                            int32_t prior, double gamma_a, double gamma_b,
                            double *w, double *h, double *lk, int32_t nthreads)
{
    if (n <= 0 || m <= 0 || r <= 0) return -1;
    int nt = 1;
#ifdef _OPENMP
    nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads;
#endif
    const size_t nr = (size_t)n * r;
    double *wr = malloc(nr * sizeof(double));                   /* row-major copy of w: a gene's r values adjacent */
    double *accp = calloc(nr * (size_t)nt, sizeof(double));     /* per-thread gene-side accumulators */
    double *cs = malloc(r * sizeof(double)), *rs = malloc(r * sizeof(double));
    double *dp = calloc((size_t)nt, sizeof(double));
    if (!wr || !accp || !cs || !rs || !dp) { free(wr); free(accp); free(cs); free(rs); free(dp); return -1; }
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) wr[i * r + k] = w_in[i + (size_t)k * n];

    /* :8-15  H update */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += w_in[i + (size_t)k * n];
        cs[k] = s;
    }
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t j = 0; j < m; j++) {
        const double *hj = h_in + (size_t)j * r;
        double acc[64];
        for (int32_t k = 0; k < r; k++) acc[k] = 0.0;
        for (int32_t e = p[j]; e < p[j + 1]; e++) {
            const double *wi = wr + (size_t)ri[e] * r;
            double wh = 0.0;
            for (int32_t k = 0; k < r; k++) wh += wi[k] * hj[k];
            const double q = x[e] / wh;
            for (int32_t k = 0; k < r; k++) acc[k] += wi[k] * q;
        }
        for (int32_t k = 0; k < r; k++) {
            double up = hj[k] * acc[k], down = cs[k];
            if (prior) { up = up + gamma_a - 1.0; down = down + gamma_a / gamma_b; }
            double v = up / down;
            if (v < ML_EPS) v = ML_EPS;
            h[k + (size_t)j * r] = v;
        }
    }
    /* :17-24  W update on the new h */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t j = 0; j < m; j++) s += h[k + (size_t)j * r];
        rs[k] = s;
    }
#pragma omp parallel num_threads(nt)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        double *acct = accp + (size_t)t * nr;
#pragma omp for schedule(static)
        for (int64_t j = 0; j < m; j++) {
            const double *hj = h + (size_t)j * r;
            for (int32_t e = p[j]; e < p[j + 1]; e++) {
                const double *wi = wr + (size_t)ri[e] * r;
                double wh = 0.0;
                for (int32_t k = 0; k < r; k++) wh += wi[k] * hj[k];
                const double q = x[e] / wh;
                double *ai = acct + (size_t)ri[e] * r;
                for (int32_t k = 0; k < r; k++) ai[k] += q * hj[k];
            }
        }
    }
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            double s = 0.0;
            for (int t = 0; t < nt; t++) s += accp[(size_t)t * nr + i * r + k];
            double up = wr[i * r + k] * s, down = rs[k];
            if (prior) { up = up + gamma_a - 1.0; down = down + gamma_a / gamma_b; }
            double v = up / down;
            if (v < ML_EPS) v = ML_EPS;
            w[i + (size_t)k * n] = v;
        }
    /* :40-49  likelihood of the updated pair */
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) wr[i * r + k] = w[i + (size_t)k * n];
#pragma omp parallel num_threads(nt)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        double s = 0.0;
#pragma omp for schedule(static)
        for (int64_t j = 0; j < m; j++) {
            const double *hj = h + (size_t)j * r;
            for (int32_t e = p[j]; e < p[j + 1]; e++) {
                const double *wi = wr + (size_t)ri[e] * r;
                double wh = 0.0;
                for (int32_t k = 0; k < r; k++) wh += wi[k] * hj[k];
                s += x[e] * log(wh);
                if (x[e] > 0.0) s += -x[e] * log(x[e]) + x[e];
            }
        }
        dp[t] = s;
    }
    double data = 0.0, cross = 0.0;
    for (int t = 0; t < nt; t++) data += dp[t];
    for (int32_t k = 0; k < r; k++) {
        double c = 0.0;
        for (int64_t i = 0; i < n; i++) c += w[i + (size_t)k * n];
        cross += c * rs[k];                                      /* rs = rowSums of the new h */
    }
    *lk = (data - cross) / (double)n / (double)m;
    free(wr); free(accp); free(cs); free(rs); free(dp);
    return 0;
}
