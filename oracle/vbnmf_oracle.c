/*
 * vbnmf_oracle.c -- CPU restatement of ccfindR's variational-Bayes NMF update step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ccfindr_amd/, the HIP
 * library, include/) may link, load or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference (hjunwoo/ccfindR v1.5.1) ships no tests, golden
 * vectors or stored outputs for this path, and its native file needs Rcpp, RcppEigen
 * (Eigen) and GSL, none of which exist in this image, so it can be neither compiled
 * nor run here.  This restatement follows the reference source statement by
 * statement (citations below) and is cross-checked against an independent numpy
 * restatement of the R twin (oracle/vbnmf_oracle.py) and against scipy/mpmath for the
 * special functions the reference takes from GSL (gsl_sf_psi, gsl_sf_lngamma; system
 * library, version unpinned by src/Makevars:2).
 *
 * Reference lines restated (relative to the reference checkout):
 *   src/vbnmf_update.cpp:19-31   prologue (dims, lw/lh/ew/eh, hyper)
 *   src/vbnmf_update.cpp:33-36   wth, xwh, sw, sh
 *   src/vbnmf_update.cpp:38-46   alw, bew, ew, dw
 *   src/vbnmf_update.cpp:48-56   alh, beh, eh, dh   (uses the NEW ew)
 *   src/vbnmf_update.cpp:58-65   lw, lh = max(exp(psi(al))/be, fudge)
 *   src/vbnmf_update.cpp:67-90   log evidence U, divided by n*m
 *   R/bayesian.R:56-106          the R twin of the same step (cross-reference)
 *
 * All matrices are column-major doubles, as R and Eigen hold them:
 *   lw, ew, dw : n x r  (element (i,k) at [i + k*n])
 *   lh, eh, dh : r x m  (element (k,j) at [k + j*r])
 *   X          : n x m  (element (i,j) at [i + j*n])
 *
 * One deliberate difference: the reference divides by n*m computed in int
 * (src/vbnmf_update.cpp:90), which overflows past 2^31-1 elements; here the
 * product is formed in double, as the R twin does (R/bayesian.R:97).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- special functions (GSL is absent; standard definitions) ------------------ */

/* psi(x), x > 0.  Upward recurrence to x >= 10, then the asymptotic series
 * ln x - 1/(2x) - sum B_2k/(2k x^2k).  Replaces gsl_sf_psi at
 * src/vbnmf_update.cpp:59,63. */
double oracle_digamma(double x)
{
    if (!(x > 0.0)) return NAN;
    double s = 0.0;
    while (x < 10.0) { s -= 1.0 / x; x += 1.0; }
    double xi = 1.0 / x, y = xi * xi;
    double ser = y * (1.0 / 12 - y * (1.0 / 120 - y * (1.0 / 252 - y * (1.0 / 240
               - y * (1.0 / 132 - y * (691.0 / 32760 - y * (1.0 / 12)))))));
    return s + log(x) - 0.5 * xi - ser;
}

/* psi'(x), x > 0 (psigamma(x, 1) of R/bayesian.R:20,24). */
double oracle_trigamma(double x)
{
    if (!(x > 0.0)) return NAN;
    double s = 0.0;
    while (x < 10.0) { s += 1.0 / (x * x); x += 1.0; }
    double xi = 1.0 / x, y = xi * xi;
    /* 1/x + 1/(2x^2) + sum B_2k / x^(2k+1) */
    double ser = xi * y * (1.0 / 6 - y * (1.0 / 30 - y * (1.0 / 42 - y * (1.0 / 30
               - y * (5.0 / 66 - y * (691.0 / 2730 - y * (7.0 / 6)))))));
    return s + xi + 0.5 * y + ser;
}

/* ln Gamma(x): libm's lgamma (glibc, < 1 ulp-class) stands in for gsl_sf_lngamma
 * at src/vbnmf_update.cpp:81,82,85,87,89. */
double oracle_lgamma(double x) { return lgamma(x); }

/* ---- the literal dense step ---------------------------------------------------- */

/* C = A(n x r) * B(r x m), column-major, plain triple loop, k innermost. */
static void gemm_nn(int64_t n, int64_t r, int64_t m, const double *A, const double *B, double *C)
{
    for (int64_t j = 0; j < m; j++)
        for (int64_t i = 0; i < n; i++) {
            double s = 0.0;
            for (int64_t k = 0; k < r; k++) s += A[i + k * n] * B[k + j * r];
            C[i + j * n] = s;
        }
}

/*
 * One call of vbnmf_update(X, wh, hyper, fudge), dense, single-threaded, same
 * statement order as src/vbnmf_update.cpp:33-90.  ew_in is read by the reference
 * (:24) but never used before being overwritten (:44), so it is not an argument.
 * Returns 0, or -1 on allocation failure / bad dims.
 */
int oracle_vbnmf_update_dense(int64_t n, int64_t m, int32_t r, const double *X,
                              const double *lw_in, const double *lh_in, const double *eh_in,
                              double aw, double bw, double ah, double bh, double fudge,
                              double *lw, double *lh, double *ew, double *eh,
                              double *dw, double *dh, double *lkh)
{
    if (n <= 0 || m <= 0 || r <= 0) return -1;
    size_t nm = (size_t)n * (size_t)m, nr = (size_t)n * r, rm = (size_t)r * m;
    double *wth = malloc(nm * sizeof(double));
    double *xwh = malloc(nm * sizeof(double));
    double *sw = malloc(nr * sizeof(double)), *sh = malloc(rm * sizeof(double));
    double *alw = malloc(nr * sizeof(double)), *alh = malloc(rm * sizeof(double));
    double *bew = malloc(r * sizeof(double)), *beh = malloc(r * sizeof(double));
    double *A = malloc(nm * sizeof(double)), *B = malloc(nm * sizeof(double));
    double *t1 = malloc(nr * sizeof(double)), *t2 = malloc(rm * sizeof(double));
    if (!wth || !xwh || !sw || !sh || !alw || !alh || !bew || !beh || !A || !B || !t1 || !t2) {
        free(wth); free(xwh); free(sw); free(sh); free(alw); free(alh);
        free(bew); free(beh); free(A); free(B); free(t1); free(t2);
        return -1;
    }

    /* :33 wth = lw*lh ; :34 xwh = X/wth */
    gemm_nn(n, r, m, lw_in, lh_in, wth);
    for (size_t e = 0; e < nm; e++) xwh[e] = X[e] / wth[e];
    /* :35 sw = lw .* (xwh * lh^T) */
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            double s = 0.0;
            for (int64_t j = 0; j < m; j++) s += xwh[i + j * n] * lh_in[k + j * r];
            sw[i + k * n] = lw_in[i + k * n] * s;
        }
    /* :36 sh = lh .* (lw^T * xwh) */
    for (int64_t j = 0; j < m; j++)
        for (int32_t k = 0; k < r; k++) {
            double s = 0.0;
            for (int64_t i = 0; i < n; i++) s += lw_in[i + k * n] * xwh[i + j * n];
            sh[k + j * r] = lh_in[k + j * r] * s;
        }

    /* :38-46 alw = aw + sw ; bew(i,k) = aw/bw + rowSums(eh)(k) ; ew ; dw */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t j = 0; j < m; j++) s += eh_in[k + j * r];
        bew[k] = aw / bw + s;
    }
    for (int32_t k = 0; k < r; k++)
        for (int64_t i = 0; i < n; i++) {
            size_t e = i + (size_t)k * n;
            alw[e] = aw + sw[e];
            ew[e] = alw[e] / bew[k];
            dw[e] = alw[e] / bew[k] / bew[k];
        }
    /* :48-56 alh = ah + sh ; beh(k,j) = ah/bh + colSums(ew_new)(k) ; eh ; dh */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += ew[i + (size_t)k * n];
        beh[k] = ah / bh + s;
    }
    for (int64_t j = 0; j < m; j++)
        for (int32_t k = 0; k < r; k++) {
            size_t e = k + (size_t)j * r;
            alh[e] = ah + sh[e];
            eh[e] = alh[e] / beh[k];
            dh[e] = alh[e] / beh[k] / beh[k];
        }
    /* :58-65 geometric means with the fudge floor */
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            size_t e = i + (size_t)k * n;
            double tmp = exp(oracle_digamma(alw[e])) / bew[k];
            lw[e] = (tmp > fudge ? tmp : fudge);
        }
    for (int32_t k = 0; k < r; k++)
        for (int64_t j = 0; j < m; j++) {
            size_t e = k + (size_t)j * r;
            double tmp = exp(oracle_digamma(alh[e])) / beh[k];
            lh[e] = (tmp > fudge ? tmp : fudge);
        }

    /* :67 wth = lw*lh (new) ; :69-72 A = (lw.*log lw)*lh ; B = lw*(lh.*log lh) */
    gemm_nn(n, r, m, lw, lh, wth);
    for (size_t e = 0; e < nr; e++) t1[e] = lw[e] * log(lw[e]);
    gemm_nn(n, r, m, t1, lh, A);
    for (size_t e = 0; e < rm; e++) t2[e] = lh[e] * log(lh[e]);
    gemm_nn(n, r, m, lw, t2, B);
    /* :73-78 U1 = -ew*eh - X .* ((A+B)/wth - log wth)   (xwh reused for ew*eh) */
    gemm_nn(n, r, m, ew, eh, xwh);
    /* :79-81 U = sum(U1 - lgamma(X+1)), i outer, j inner as the reference loops */
    double U = 0.0;
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++) {
            size_t e = i + (size_t)j * n;
            double u1 = (A[e] + B[e]) / wth[e];
            u1 = u1 - log(wth[e]);
            u1 = X[e] * u1;
            u1 = -xwh[e] - u1;
            U += u1 - oracle_lgamma(X[e] + 1.0);
        }
    /* :82-86 */
    double lga = -oracle_lgamma(aw) + aw * log(aw / bw);
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            size_t e = i + (size_t)k * n;
            U += -(aw / bw) * ew[e] + lga + alw[e] * (1.0 - log(bew[k])) + oracle_lgamma(alw[e]);
        }
    /* :87-89 */
    lga = -oracle_lgamma(ah) + ah * log(ah / bh);
    for (int32_t k = 0; k < r; k++)
        for (int64_t j = 0; j < m; j++) {
            size_t e = k + (size_t)j * r;
            U += -(ah / bh) * eh[e] + lga + alh[e] * (1.0 - log(beh[k])) + oracle_lgamma(alh[e]);
        }
    /* :90 (in double, see header) */
    U /= (double)n * (double)m;
    *lkh = U;

    free(wth); free(xwh); free(sw); free(sh); free(alw); free(alh);
    free(bew); free(beh); free(A); free(B); free(t1); free(t2);
    return 0;
}

/* ---- the same step with X held sparse (CSC, dgCMatrix slots) -------------------- */

/*
 * Same mathematics with X in compressed-sparse-column form (p: m+1 column pointers,
 * i: row indices, x: values), visiting stored entries only: an absent entry has
 * X_ij = 0, contributes 0 to sw, sh (:34-36) and to X.*(...) (:77), lgamma(0+1) = 0
 * (:81), and sum_ij (ew*eh)_ij = sum_k colSum(ew)_k rowSum(eh)_k (:78).  The data
 * term is evaluated per entry exactly as :69-77 write it, (A+B)/wth - log wth.
 * OpenMP over columns when compiled with -fopenmp (nthreads <= 0: library default);
 * partial sums are combined in thread order, so the result depends on the thread
 * count in the last bits only.
 */
int oracle_vbnmf_update_csc(int64_t n, int64_t m, int32_t r,
                            const int32_t *p, const int32_t *ri, const double *x,
                            const double *lw_in, const double *lh_in, const double *eh_in,
                            double aw, double bw, double ah, double bh, double fudge,
                            double *lw, double *lh, double *ew, double *eh,
                            double *dw, double *dh, double *lkh, int32_t nthreads)
{
    if (n <= 0 || m <= 0 || r <= 0) return -1;
    int nt = 1;
#ifdef _OPENMP
    nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads;
#endif
    size_t nr = (size_t)n * r, rm = (size_t)r * m;
    /* row-major copies of the gene-side factors so a gene's r values are adjacent */
    double *lwr = malloc(nr * sizeof(double));
    double *swp = calloc(nr * (size_t)nt, sizeof(double));   /* per-thread sw accumulators [t][i][k] */
    double *sh = malloc(rm * sizeof(double));
    double *alw = malloc(nr * sizeof(double)), *alh = malloc(rm * sizeof(double));
    double *bew = malloc(r * sizeof(double)), *beh = malloc(r * sizeof(double));
    double *Up = calloc((size_t)nt, sizeof(double));
    if (!lwr || !swp || !sh || !alw || !alh || !bew || !beh || !Up) {
        free(lwr); free(swp); free(sh); free(alw); free(alh); free(bew); free(beh); free(Up);
        return -1;
    }
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) lwr[i * r + k] = lw_in[i + (size_t)k * n];

    /* :33-36 on stored entries */
#pragma omp parallel num_threads(nt)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        double *swt = swp + (size_t)t * nr;
        double *acc = malloc(r * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t j = 0; j < m; j++) {
            const double *lhj = lh_in + (size_t)j * r;
            for (int32_t k = 0; k < r; k++) acc[k] = 0.0;
            for (int32_t e = p[j]; e < p[j + 1]; e++) {
                const double *lwi = lwr + (size_t)ri[e] * r;
                double w = 0.0;
                for (int32_t k = 0; k < r; k++) w += lwi[k] * lhj[k];
                double q = x[e] / w;
                double *swi = swt + (size_t)ri[e] * r;
                for (int32_t k = 0; k < r; k++) { swi[k] += q * lhj[k]; acc[k] += lwi[k] * q; }
            }
            for (int32_t k = 0; k < r; k++) sh[k + (size_t)j * r] = lhj[k] * acc[k];
        }
        free(acc);
    }
    /* :38-46 */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t j = 0; j < m; j++) s += eh_in[k + (size_t)j * r];
        bew[k] = aw / bw + s;
    }
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            double s = 0.0;
            for (int t = 0; t < nt; t++) s += swp[(size_t)t * nr + i * r + k];
            size_t e = i + (size_t)k * n;
            alw[e] = aw + lw_in[e] * s;
            ew[e] = alw[e] / bew[k];
            dw[e] = alw[e] / bew[k] / bew[k];
        }
    /* :48-56 */
    for (int32_t k = 0; k < r; k++) {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += ew[i + (size_t)k * n];
        beh[k] = ah / bh + s;
    }
    for (int64_t j = 0; j < m; j++)
        for (int32_t k = 0; k < r; k++) {
            size_t e = k + (size_t)j * r;
            alh[e] = ah + sh[e];
            eh[e] = alh[e] / beh[k];
            dh[e] = alh[e] / beh[k] / beh[k];
        }
    /* :58-65 */
    for (size_t e = 0; e < nr; e++) {
        double tmp = exp(oracle_digamma(alw[e])) / bew[e / n];
        lw[e] = (tmp > fudge ? tmp : fudge);
    }
    for (size_t e = 0; e < rm; e++) {
        double tmp = exp(oracle_digamma(alh[e])) / beh[e % r];
        lh[e] = (tmp > fudge ? tmp : fudge);
    }
    /* :67-81 on stored entries, per-entry (A+B)/wth - log wth */
    double *llwr = swp;                       /* reuse: [i][k] = lw*log(lw), row-major */
    for (int64_t i = 0; i < n; i++)
        for (int32_t k = 0; k < r; k++) {
            double v = lw[i + (size_t)k * n];
            lwr[i * r + k] = v;
            llwr[i * r + k] = v * log(v);
        }
#pragma omp parallel num_threads(nt)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        double *llh = malloc(r * sizeof(double));
        double Ut = 0.0;
#pragma omp for schedule(static)
        for (int64_t j = 0; j < m; j++) {
            const double *lhj = lh + (size_t)j * r;
            for (int32_t k = 0; k < r; k++) llh[k] = lhj[k] * log(lhj[k]);
            for (int32_t e = p[j]; e < p[j + 1]; e++) {
                const double *lwi = lwr + (size_t)ri[e] * r;
                const double *llwi = llwr + (size_t)ri[e] * r;
                double w = 0.0, a = 0.0, b = 0.0;
                for (int32_t k = 0; k < r; k++) {
                    w += lwi[k] * lhj[k];
                    a += llwi[k] * lhj[k];
                    b += lwi[k] * llh[k];
                }
                Ut += -x[e] * ((a + b) / w - log(w)) - oracle_lgamma(x[e] + 1.0);
            }
        }
        Up[t] = Ut;
        free(llh);
    }
    double U = 0.0;
    for (int t = 0; t < nt; t++) U += Up[t];
    /* :78 -sum(ew*eh) collapsed */
    for (int32_t k = 0; k < r; k++) {
        double cw = 0.0, rh = 0.0;
        for (int64_t i = 0; i < n; i++) cw += ew[i + (size_t)k * n];
        for (int64_t j = 0; j < m; j++) rh += eh[k + (size_t)j * r];
        U -= cw * rh;
    }
    /* :82-89 */
    double lga = -oracle_lgamma(aw) + aw * log(aw / bw);
    double Uw = 0.0, Uh = 0.0;
    for (size_t e = 0; e < nr; e++)
        Uw += -(aw / bw) * ew[e] + lga + alw[e] * (1.0 - log(bew[e / n])) + oracle_lgamma(alw[e]);
    lga = -oracle_lgamma(ah) + ah * log(ah / bh);
    for (size_t e = 0; e < rm; e++)
        Uh += -(ah / bh) * eh[e] + lga + alh[e] * (1.0 - log(beh[e % r])) + oracle_lgamma(alh[e]);
    U += Uw + Uh;
    U /= (double)n * (double)m;
    *lkh = U;

    free(lwr); free(swp); free(sh); free(alw); free(alh); free(bew); free(beh); free(Up);
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
