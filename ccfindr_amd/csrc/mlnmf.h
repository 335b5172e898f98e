// mlnmf.h -- gfx950 kernels of the maximum-likelihood NMF step (included once, by engine.hip, after kernels.h).
//
// Reference: R/factorize.R:2-27 (nmf_updateR) and :40-49 (likelihood), behind factorize() (:140-320).
//   :8-15   h <- h .* (t(w) %*% (x / (w %*% h))) / colSums(w)   [+ Gamma prior: up + a - 1, down + a/b] ; clip at eps
//   :17-24  w <- w .* ((x / (w %*% h_new)) %*% t(h_new)) / rowSums(h_new)                               ; clip at eps
//   :40-49  lk = ( sum(x log(wh) - wh) + sum_{x>0}(-x log x + x) ) / n / m    on the NEW w, h
// The two updates are sequential (the W update sees the new h), so a step makes two single-side sweeps over
// X (k_sweep1 in kernels.h, same tiled layout and per-task partials as the VB sweep):
//   k_ml_update(H) -> k_sweep1(gene side: w, h_new) -> k_ml_update(W) -> k_sweep1(cell side: h_new, w_new) -> k_ml_final
// The cell-side sweep at the END of step t runs on (w_new, h_new): it yields the statistics the H update of
// step t+1 starts from AND sum x log(wh) of step t's likelihood; sum(wh) = sum_k colSum(w)_k rowSum(h)_k comes
// from the updates' block partials, and sum_{x>0}(-x log x + x) is a constant of X computed once at ingestion.
#pragma once
#include "kernels.h"

namespace vbnmf {

// One factor's multiplicative update.  Thread (row_sub, k) walks its block's majors with a fixed k:
//   s    = sum of the major's task partials (fixed order)          = (t(w) %*% (x/wh))[k, major] or its W twin
//   up   = f * s [+ a - 1] ; down = colsum_other[k] [+ a/b] ; f <- max(up / down, eps)   (NaN stays NaN, as in R)
// other_bp[other_nb][R+2] are the OTHER factor's block partials of its column sums; this factor's go to bp.
// The control step of the device-driven ML loop (k_ml_control below) folded into the H update of the NEXT step, exactly as
// the VB loop folds its own into the gene-side update (kernels.h: ControlFold): every block forms it from the same
// inputs, block 0 writes it out, and what a block reads is never written in the same launch (the control block and the
// table of H-side block partials alternate between two buffers by step parity).
struct MlFold {
    const LoopCtl *prev;           // null: no fold
    LoopCtl *next;
    const double *bpH_prev;        // [nb][R+2] block partials of the previous step's H update
    const double *epart;           // the previous cell-side sweep's sum x log(wh) partials
    int64_t nepart;
    double xlx, n, m;
    double *history, *out_host;
    int32_t do_control, control_only;
};

template <int R>
__device__ __forceinline__ void ml_update_body(
    const double *__restrict__ part, const int32_t *__restrict__ inv_ptr, const uint32_t *__restrict__ inv_task,
    int64_t nmaj, int r, const double *__restrict__ other_bp, int other_nb, int prior, double ga, double gb, double eps,
    double *__restrict__ f, double *__restrict__ bp, const int32_t *__restrict__ stop, const MlFold &fold, int stage_ids)
{
    constexpr int RB = kUpdateThreads / R;       // majors per pass
    __shared__ double s_other[R + 2];
    __shared__ double s_e[kUpdateThreads];
    __shared__ uint32_t s_ids[kStageIds];
    __shared__ int32_t s_ptr[kStagePtr];
    const int t = threadIdx.x;
    // the block's stretch of the inverse index into LDS, first thing (as k_update, kernels.h)
    const int64_t per0 = (nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t bm0 = (int64_t)blockIdx.x * per0, bm1 = min(nmaj, bm0 + per0);
    int q_lo = 0;
    bool staged = false;
    if (stage_ids && !fold.control_only && bm0 < bm1 && bm1 - bm0 < kStagePtr) {
        q_lo = inv_ptr[bm0];
        const int q_hi = inv_ptr[bm1];
        staged = q_hi - q_lo <= kStageIds;               // (block-uniform)
        if (staged) {
            for (int q = t; q <= (int)(bm1 - bm0); q += kUpdateThreads) s_ptr[q] = inv_ptr[bm0 + q];
            for (int q = q_lo + t; q < q_hi; q += kUpdateThreads) s_ids[q - q_lo] = inv_task[q];
        }
    }
    if (fold.prev) {
        // ---- the folded control step: k_ml_control's arithmetic, statement by statement ----
        __shared__ double sH[R + 2];
        __shared__ int s_stop;
        const LoopCtl *pv = fold.prev;
        const int was_stopped = pv->stop;
        double pe = 0.0;
        if (fold.do_control) for (int64_t q = t; q < fold.nepart; q += kUpdateThreads) pe += fold.epart[q];
        bp_colsums2(other_bp, fold.bpH_prev, other_nb, R + 2, s_other, sH, kUpdateThreads);   // colSums(w) (this update's `down`), colSums(h)
        if (was_stopped) {                       // a step queued past the stop: everything travels on unchanged
            if (blockIdx.x == 0 && t == 0) *fold.next = *pv;
            if (t < R + 2 && !fold.control_only) bp[(size_t)blockIdx.x * (R + 2) + t] = fold.bpH_prev[(size_t)blockIdx.x * (R + 2) + t];
            return;
        }
        const double data = block_sum(pe, s_e);
        if (t == 0) {
            int reason = 0, it = pv->it;
            double lk = pv->lkh, new_lk0 = pv->lk0;
            if (fold.do_control) {
                double cross = 0.0;
                for (int k = 0; k < r; k++) cross += s_other[k] * sH[k];
                lk = ((data - cross) + fold.xlx) / fold.n / fold.m;
                it = pv->it + 1;
                const double lkold = pv->lk0;
                if (fabs(lkold - lk) < pv->tol * fabs(lkold)) reason = 2;           // converged (R/factorize.R:211)
                else { new_lk0 = lk; if (it >= pv->max_it) reason = 4; }
            }
            s_stop = reason != 0;
            if (blockIdx.x == 0) {
                LoopCtl nx = *pv;
                nx.it = it; nx.lkh = lk; nx.lk0 = new_lk0;
                if (reason) { nx.reason = reason; nx.stop = 1; }
                *fold.next = nx;
                if (fold.do_control) {
                    if (fold.history) fold.history[it - 1] = lk;
                    double *oh = fold.out_host;
                    oh[0] = lk;
                    oh[12] = new_lk0;
                    oh[5] = (double)it;
                    __threadfence_system();
                    reinterpret_cast<volatile double *>(oh)[6] = (double)reason;
                    reinterpret_cast<volatile double *>(oh)[7] = (double)it;
                }
            }
        }
        __syncthreads();
        if (fold.control_only) return;
        if (s_stop) {
            if (t < R + 2) bp[(size_t)blockIdx.x * (R + 2) + t] = fold.bpH_prev[(size_t)blockIdx.x * (R + 2) + t];
            return;
        }
    } else {
    const int stopped = stop ? *stop : 0;        // device-driven loop: the run has ended, leave the factors as they are
    bp_colsums(other_bp, other_nb, R + 2, s_other, kUpdateThreads);       // (its loads travel with the flag's)
    if (stopped) return;
    __syncthreads();
    }

    const int row = t / R, k = t - row * R;
    const int64_t per = (nmaj + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = min(nmaj, m0 + per);
    double down = s_other[k < R ? k : 0];
    if (prior) down = down + ga / gb;            // R/factorize.R:12,21
    double ve = 0.0;
    if (row < RB) {
        for (int64_t M = m0 + row; M < m1; M += RB) {
            const size_t o = (size_t)M * R + k;
            if (k < r) {
                const double s = staged ? task_sum_lds(part, s_ids, s_ptr[M - bm0] - q_lo, s_ptr[M - bm0 + 1] - q_lo, R, k)
                                        : task_sum(part, inv_task, inv_ptr[M], inv_ptr[M + 1], R, k);
                double up = f[o] * s;
                if (prior) up = up + ga - 1.0;   // :11,20
                double v = up / down;
                if (v < eps) v = eps;            // :15,24
                f[o] = v;
                ve += v;
            } else {
                f[o] = 0.0;
            }
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

template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_ml_update(
    const double *__restrict__ part, const int32_t *__restrict__ inv_ptr, const uint32_t *__restrict__ inv_task,
    int64_t nmaj, int r, const double *__restrict__ other_bp, int other_nb, int prior, double ga, double gb, double eps,
    double *__restrict__ f, double *__restrict__ bp, const int32_t *__restrict__ stop, const MlFold fold, int stage_ids)
{
    ml_update_body<R>(part, inv_ptr, inv_task, nmaj, r, other_bp, other_nb, prior, ga, gb, eps, f, bp, stop, fold, stage_ids);
}

// A BATCH of engines stepped by one launch (kernels.h: k_update2_batch; here the restarts of factorize(), reference
// R/factorize.R:181: `for(irun in seq_len(nrun))`): blockIdx.y picks the engine's argument block, the body is the single engine's.
struct MlUpdJob {
    const double *part;
    const int32_t *inv_ptr;
    const uint32_t *inv_task;
    int64_t nmaj;
    const double *other_bp;
    double *f, *bp;
    const int32_t *stop;
    double ga, gb, eps;
    int32_t r, other_nb, prior, stage_ids;
    MlFold fold;
};

template <int R>
__global__ __launch_bounds__(kUpdateThreads) void k_ml_update_batch(const MlUpdJob *__restrict__ jobs)
{
    const MlUpdJob J = jobs[blockIdx.y];
    ml_update_body<R>(J.part, J.inv_ptr, J.inv_task, J.nmaj, J.r, J.other_bp, J.other_nb, J.prior, J.ga, J.gb, J.eps, J.f, J.bp, J.stop,
                      J.fold, J.stage_ids);
}

template <int R, bool WIDE, bool LOGTERM, int NT>
__global__ __launch_bounds__(NT) void k_sweep1_batch(const SweepSide *__restrict__ jobs)
{
    extern __shared__ double2 ldsG[];
    const SweepSide S = jobs[blockIdx.y];
    if (S.stop && *S.stop) return;
    sweep_side<R, WIDE, LOGTERM, NT, LOGTERM ? 2 : 0, 1>(S, ldsG);
}

// Likelihood (R/factorize.R:40-49).  One block.  out = [lk, sum x log(wh), sum(wh), 0, 0], out_host[7] = seq.
template <int R>
__global__ __launch_bounds__(1024) void k_ml_final(const double *__restrict__ bpW, const double *__restrict__ bpH, int nb,
                                                   const double *__restrict__ epart, int64_t nepart, double xlx, int r,
                                                   double n, double m, double seq, double *__restrict__ out,
                                                   double *__restrict__ out_host)
{
    __shared__ double sW[R + 2], sH[R + 2];
    __shared__ double sm[1024];
    double part = 0.0;                           // the three reductions' loads travel together
    for (int64_t q = threadIdx.x; q < nepart; q += 1024) part += epart[q];
    bp_colsums2(bpW, bpH, nb, R + 2, sW, sH, 1024);
    const double data = block_sum(part, sm);
    if (threadIdx.x == 0) {
        double cross = 0.0;
        for (int k = 0; k < r; k++) cross += sW[k] * sH[k];
        double o[5] = {((data - cross) + xlx) / n / m, data, cross, 0.0, 0.0};
        for (int q = 0; q < 5; q++) { out[q] = o[q]; out_host[q] = o[q]; }
        __threadfence_system();
        reinterpret_cast<volatile double *>(out_host)[7] = seq;
    }
}

// Device-driven loop of factorize() under criterion = 'likelihood' (R/factorize.R:194-213): after each step the
// likelihood, then `if (abs(lkold - lk0) < Tol * abs(lkold)) break ; lkold <- lk0` (:211-213; lkold starts at -Inf, so
// the first pass never breaks; a NaN likelihood never satisfies the test either, as in R's host loop mirror).
// LoopCtl: lk0 holds lkold, lkh the last likelihood.  history[it-1] = lk.  out_host = [lk, ., ., ., ., it, reason, it].
template <int R>
__global__ __launch_bounds__(1024) void k_ml_control(const double *__restrict__ bpW, const double *__restrict__ bpH, int nb,
                                                     const double *__restrict__ epart, int64_t nepart, double xlx, int r,
                                                     double n, double m, LoopCtl *ctl, double *__restrict__ history,
                                                     double *__restrict__ out_host)
{
    __shared__ double sW[R + 2], sH[R + 2];
    __shared__ double sm[1024];
    const int stopped = ctl->stop;               // tested once the reductions' loads are in flight too
    double part = 0.0;
    for (int64_t q = threadIdx.x; q < nepart; q += 1024) part += epart[q];
    bp_colsums2(bpW, bpH, nb, R + 2, sW, sH, 1024);
    if (stopped) return;
    const double data = block_sum(part, sm);
    if (threadIdx.x != 0) return;
    double cross = 0.0;
    for (int k = 0; k < r; k++) cross += sW[k] * sH[k];
    const double lk = ((data - cross) + xlx) / n / m;
    const int it = ctl->it + 1;
    const double lkold = ctl->lk0;
    int reason = 0;
    if (fabs(lkold - lk) < ctl->tol * fabs(lkold)) reason = 2;           // converged (:211)
    else { ctl->lk0 = lk; if (it >= ctl->max_it) reason = 4; }
    ctl->it = it; ctl->lkh = lk;
    if (history) history[it - 1] = lk;
    if (reason) { ctl->reason = reason; ctl->stop = 1; }
    out_host[0] = lk;
    out_host[12] = ctl->lk0;
    out_host[5] = (double)it;
    __threadfence_system();
    reinterpret_cast<volatile double *>(out_host)[6] = (double)reason;
    reinterpret_cast<volatile double *>(out_host)[7] = (double)it;
}

// which.max(h[, j])[1] for every cell j (reference R/factorize.R:55-56, R/utils.R:906): 1-based index of the first
// maximum among the r components.  NaN entries never win (R's which.max skips them); a column of NaNs gives 0.
__global__ __launch_bounds__(256) void k_argmax(const double *__restrict__ h, int64_t m, int r, int R, int32_t *__restrict__ ids)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double *row = h + (size_t)j * R;
    int best = 0;
    double bv = 0.0;
    for (int k = 0; k < r; k++) {
        const double v = row[k];
        if (v == v && (best == 0 || v > bv)) { best = k + 1; bv = v; }
    }
    ids[j] = best;
}

}  // namespace vbnmf
