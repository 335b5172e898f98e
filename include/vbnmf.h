/*
 * vbnmf.h -- C ABI of the MI355X-native VB-NMF update engine (libvbnmf_hip.so).
 *
 * Drop-in boundary for ONE path of ccfindR: the variational-Bayes NMF update step
 *     Rcpp::List vbnmf_update(const Eigen::MatrixXd& X, const Rcpp::List& wh,
 *                             const Rcpp::List& hyper, const Rcpp::NumericVector& fudge)
 * (reference src/vbnmf_update.cpp:16-17), reached from R through
 *     .Call(`_ccfindR_vbnmf_update`, X, wh, hyper, fudge)
 * (reference R/RcppExports.R:4-6, src/RcppExports.cpp:11-22) and called only by
 * vb_iterate (reference R/bayesian.R:339).  INTEGRATION.md shows the Rcpp shim that
 * binds these entry points under the reference's own .Call symbol.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns a vbnmf_status (0 = ok) and
 *     never throws or aborts; vbnmf_last_error() gives the message of the calling
 *     thread's last failure (the shim turns it into Rcpp::stop, as END_RCPP does for
 *     exceptions at src/RcppExports.cpp:21).
 *   - matrices are column-major fp64 exactly as R / Eigen hold them:
 *       lw, ew, dw : n x r        lh, eh, dh : r x m        X : n x m
 *   - dimensions are 64-bit (the reference's `U/=n*m` in int, src/vbnmf_update.cpp:90,
 *     overflows past 2^31-1 elements; here the product is formed in double as the R twin
 *     does, R/bayesian.R:97).
 *   - NaN/Inf are not errors: they propagate into lkh, so the caller's is.na() break
 *     (R/bayesian.R:345) keeps working.
 *   - a HIP device is REQUIRED.  There is no CPU fallback: without a usable gfx950
 *     device every compute entry point fails with VBNMF_ERR_NO_DEVICE.
 *   - one engine = one device = one host thread at a time; distinct engines may be
 *     driven from distinct threads / processes (one process per GPU).
 */
#ifndef VBNMF_H
#define VBNMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VBNMF_OK = 0,
    VBNMF_ERR_BAD_ARG = 1,     /* null pointer, non-positive dim, rank out of range, bad index */
    VBNMF_ERR_NO_DEVICE = 2,   /* no HIP device / device is not usable                          */
    VBNMF_ERR_HIP = 3,         /* a HIP runtime call failed (message has the HIP error string)  */
    VBNMF_ERR_OOM = 4,         /* host or device allocation failed                              */
    VBNMF_ERR_STATE = 5        /* call sequence error (e.g. step before set_state)              */
} vbnmf_status;

/* Largest rank an engine takes.  The reference's only bound is rank <= min(n, m) (R/bayesian.R:319-320); here the
 * factor rows live in registers: up to 32 columns per lane, above that two lanes share a task (ranks 33..64, padded
 * to a multiple of 8) and then four (ranks 65..128, padded to a multiple of 16).  The device-resident truncated SVD
 * (vbnmf_engine_svd) holds subspaces of at most VBNMF_MAX_SVD_COLUMNS columns. */
#define VBNMF_MAX_RANK 128
#define VBNMF_MAX_SVD_COLUMNS 64

/* Message of the calling thread's most recent failure ("" if none). Never NULL. */
const char *vbnmf_last_error(void);
/* "major.minor.patch" of this library. */
const char *vbnmf_version(void);
/* Number of visible HIP devices (0 if none / no driver); never fails. */
int32_t vbnmf_device_count(void);

/* ---------------------------------------------------------------------------------
 * Count matrix X.  Host-side canonical copy (compressed sparse columns, zeros dropped)
 * plus the iteration-invariant sum_ij lgamma(X_ij + 1) of src/vbnmf_update.cpp:80-81.
 * Replaces the per-iteration `as.matrix(bundle$mat)` + Rcpp::as<Eigen::MatrixXd> copy
 * (R/bayesian.R:339, src/RcppExports.cpp:15): X is ingested ONCE.
 * --------------------------------------------------------------------------------- */
typedef struct vbnmf_matrix vbnmf_matrix;

/* Dense n x m column-major doubles (what as.matrix() hands the reference). */
int vbnmf_matrix_from_dense(int64_t n, int64_t m, const double *X, vbnmf_matrix **out);
/* dgCMatrix slots (Matrix package; what counts(object) is after read_10x, R/utils.R:34):
 * p = m+1 column pointers, i = 0-based row indices (any order within a column,
 * duplicates are summed), x = values. */
int vbnmf_matrix_from_csc(int64_t n, int64_t m, const int32_t *p, const int32_t *i,
                          const double *x, vbnmf_matrix **out);
/* Compressed sparse rows: p = n+1 row pointers, j = 0-based column indices. */
int vbnmf_matrix_from_csr(int64_t n, int64_t m, const int32_t *p, const int32_t *j,
                          const double *x, vbnmf_matrix **out);
/* Matrix Market file: what read_10x hands the reference through
 * as(Matrix::readMM(count), 'dgCMatrix') (R/utils.R:34; e.g. inst/extdata/matrix.mtx:1-2).
 * Coordinate real / integer / pattern, general / symmetric / skew-symmetric, or the dense
 * array form; 1-based indices; repeated (i, j) are summed; explicit zeros are dropped.
 * Parsed by all host threads straight into compressed columns. */
int vbnmf_matrix_from_mtx(const char *path, vbnmf_matrix **out);
/* The file Matrix::writeMM writes in write_10x (R/utils.R:876): coordinate, general,
 * "integer" when every stored value is one (else "real"), entries by columns. */
int vbnmf_matrix_write_mtx(const vbnmf_matrix *X, const char *path);
/* Read-only view of the canonical compressed columns (m+1 pointers, rows ascending in each
 * column, no zeros); the arrays belong to the handle. */
int vbnmf_matrix_csc(const vbnmf_matrix *X, const int64_t **colptr, const int32_t **row,
                     const double **val);
/* n, m, stored entries, sum lgamma(x+1); any out pointer may be NULL. */
int vbnmf_matrix_info(const vbnmf_matrix *X, int64_t *n, int64_t *m, int64_t *nnz,
                      double *sum_lgamma_x1);
/* The reference's input guards (R/bayesian.R:244-247): counts of all-zero rows / columns. */
int vbnmf_matrix_empty_counts(const vbnmf_matrix *X, int64_t *empty_rows, int64_t *empty_cols);
/* Rank classes for a sweep over several ranks on this matrix (the reference's `for(rank in ranks)`, R/bayesian.R:316,
 * which re-densifies X for every rank and iteration).  The tiled layout depends on the rank only through the LDS row
 * size; with a plan every engine created afterwards takes the geometry of the smallest class at or above its padded
 * rank, so the ranks of the sweep share ONE pair of layouts (max_classes = 1: the class of the largest rank) instead of
 * cutting one per row size.  Results do not depend on the plan beyond the summation order inside a step.
 * count = 0 clears the plan (every rank its own geometry, the default). */
int vbnmf_matrix_plan_ranks(vbnmf_matrix *X, const int32_t *ranks, int32_t count, int32_t max_classes);
/* The same classes without touching a matrix handle: classes[0..*n_classes) = padded ranks, ascending (at most `count`
 * of them).  vb_factorize picks each engine's geometry from this list and hands it to vbnmf_engine_create_geom, so a
 * sweep neither mutates the shared matrix handle nor loses a plan the caller set on it.  vbnmf_padded_rank: the rank as
 * the device pads it (even up to 32, multiples of 8 up to 64, of 16 up to 128); 0 for a rank out of range. */
int vbnmf_plan_classes(const int32_t *ranks, int32_t count, int32_t max_classes, int32_t *classes, int32_t *n_classes);
int32_t vbnmf_padded_rank(int32_t r);
/* Host threads of the library's ingestion and layout cuts (the reference is single-threaded: src/vbnmf_update.cpp has
 * no host parallelism at all).  Default: the cores of the process's affinity mask, at most 32 (VBNMF_HOST_THREADS
 * overrides) -- a cap chosen for ranks that all work at once.  vbnmf_set_host_threads(n) lifts or lowers it for THIS
 * process until called again with 0 and returns the count in force before: the one process of a node that cuts the
 * layouts for its waiting peers (vb_factorize_sharded) takes their cores for the duration of the cut.  Results do not
 * depend on the count (layouts, cell order: fixed chunking; the sums of ingestion are taken once, by the holder of X). */
int32_t vbnmf_host_threads(void);
int32_t vbnmf_set_host_threads(int32_t n);
void vbnmf_matrix_destroy(vbnmf_matrix *X);

/* ---------------------------------------------------------------------------------
 * One ingestion and one pair of tiled layouts per NODE instead of per process.  The reference ships the whole bundle,
 * matrix included, to every MPI slave (R/bayesian.R:252-263); here the processes of a node (one per GPU) share the host's
 * work through memory the caller maps into all of them (e.g. /dev/shm):
 *   builder  : vbnmf_matrix_get_meta -> meta[8]; vbnmf_matrix_export_layout(buf = NULL) cuts (and caches) the layout of one
 *              side and returns the blob size; a second call writes the blob into the shared buffer.
 *   the rest : vbnmf_matrix_shell(meta) -- a handle with X's dimensions and constants but NO entries -- then
 *              vbnmf_matrix_import_layout for each side; engines created on it (whole matrix, that geometry) upload the
 *              imported arrays to their own GPU.  Entry points that need entries (partition engines, _csc, _write_mtx,
 *              _empty_counts, vbnmf_layout_build) answer VBNMF_ERR_STATE on a shell.
 * meta = {n, m, stored entries, all-integer flag, packed-count flag, max value, sum lgamma(x+1), sum(-x log x + x)}.
 * geometry_rank and n_wg must be what the engines will use (vbnmf_engine_create_geom, vbnmf_device_sweep_workgroups).
 * --------------------------------------------------------------------------------- */
int vbnmf_matrix_get_meta(const vbnmf_matrix *X, double *meta);
int vbnmf_matrix_shell(const double *meta, vbnmf_matrix **out);
int vbnmf_matrix_is_shell(const vbnmf_matrix *X);
/* The per-matrix work every whole-matrix layout starts from, ahead of need and safe to call from a second host thread
 * while a layout is being cut: the internal order of the cells and the row-major copy of X (the gene side's input). */
int vbnmf_matrix_prepare(const vbnmf_matrix *X);
/* The same, started on a background host thread the handle owns: returns at once.  Meant for the caller that ingests X
 * for whole-matrix factorisations (vb_factorize, a rank sweep) and has other set-up to do before the first engine. */
int vbnmf_matrix_prepare_async(const vbnmf_matrix *X);
/* Uploads the whole-matrix layout of `side` (geometry_rank, n_wg as for vbnmf_matrix_export_layout; cut now if it is not
 * cached or imported yet) to HIP device `device` ahead of the first engine: engines created afterwards in that geometry
 * on that device share the resident copy (they always do among themselves; this only moves the upload earlier, e.g.
 * beside the wait for the other side's layout). */
int vbnmf_matrix_preload_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg, int32_t device);
int vbnmf_matrix_export_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg,
                               void *buf, int64_t capacity, int64_t *bytes);
int vbnmf_matrix_import_layout(const vbnmf_matrix *X, const void *buf, int64_t bytes);
/* The same without the two copies: ONE copy of a layout's big arrays (the packed entry stream, 200 MB a side at the
 * headline size) per node, in a file on a memory file system that every process maps.
 *   share  : cuts the layout with the entry stream written straight INTO the new file `path` (built as path + ".part" and
 *            renamed when complete: a peer that sees `path` sees all of it); the mapping stays the layout's storage in
 *            this process.  (A layout already cached in ordinary memory is copied into the file instead.)
 *   attach : maps `path` read-only and adopts the entry stream in place; the small index arrays are copied.
 * The file can be unlinked once every process has attached; the memory lives as long as a mapping does. */
int vbnmf_matrix_share_layout(const vbnmf_matrix *X, int32_t side, int32_t geometry_rank, int32_t n_wg, const char *path);
int vbnmf_matrix_attach_layout(const vbnmf_matrix *X, const char *path);
/* Persistent workgroups of the sweep kernels on `device` (one per CU; VBNMF_NWG overrides): the n_wg of the layouts
 * that whole-matrix engines on that device use. */
int vbnmf_device_sweep_workgroups(int32_t device, int32_t *n_wg);
/* First use of `device` by this process ahead of need (HIP context, first allocation, load of the library's code object:
 * ~0.15 s once per process): a process that must wait for something else first spends the wait here. */
int vbnmf_device_warmup(int32_t device);

/* ---------------------------------------------------------------------------------
 * Engine: device-resident state of one factorisation of (a column block of) X at one
 * rank.  Holds what vb_iterate carries between calls, {lw, lh, ew, eh} (R/bayesian.R:
 * 334-339), in HBM, plus X in the tiled device layout.
 * --------------------------------------------------------------------------------- */
typedef struct vbnmf_engine vbnmf_engine;

/* Whole matrix on HIP device `device`, rank 1 <= r <= VBNMF_MAX_RANK. */
int vbnmf_engine_create(const vbnmf_matrix *X, int32_t r, int32_t device, vbnmf_engine **out);
/* Cell-partitioned engine: owns columns [col_begin, col_end) of X; m_global = total
 * number of cells across all partitions (used for the n*m normalisation of lkh).
 * The gene-side state (lw, ew, dw) is replicated in every partition; the caller sums
 * the reduce buffer (below) across partitions between step_local and step_finish. */
int vbnmf_engine_create_part(const vbnmf_matrix *X, int64_t col_begin, int64_t col_end,
                             int64_t m_global, int32_t r, int32_t device, vbnmf_engine **out);
/* The same with the layouts' geometry named per ENGINE: geometry_rank >= r is the rank whose LDS row size the tiled
 * layouts are cut for (the ranks of a sweep, R/bayesian.R:316, share one pair: vbnmf_plan_classes); 0 = the class of the
 * matrix's plan (vbnmf_matrix_plan_ranks), the rank's own geometry without one. */
int vbnmf_engine_create_geom(const vbnmf_matrix *X, int64_t col_begin, int64_t col_end, int64_t m_global, int32_t r,
                             int32_t geometry_rank, int32_t device, vbnmf_engine **out);
void vbnmf_engine_destroy(vbnmf_engine *e);

/* n, local m (cells owned), r. */
int vbnmf_engine_dims(const vbnmf_engine *e, int64_t *n, int64_t *m_local, int32_t *r);

/* Load wh$lw (n x r), wh$lh (r x m_local), wh$eh (r x m_local): the three members of
 * `wh` that influence the result (src/vbnmf_update.cpp:22-25; ew is overwritten at :44).
 * Also (re)computes the sufficient statistics of the new state. */
int vbnmf_engine_set_state(vbnmf_engine *e, const double *lw, const double *lh, const double *eh);

/* One vbnmf_update step (src/vbnmf_update.cpp:33-90) on the resident state with
 * hyper = {aw, bw, ah, bh} and fudge.  Outputs:
 *   lkh      wh$lkh, the log evidence per element (:90)
 *   stats[4] mean(log lw), mean(log lh), mean(ew), mean(eh) of the NEW state: the four
 *            reductions hyper_update needs (R/bayesian.R:8-11), so the caller never
 *            downloads the factors between steps.  May be NULL.
 * Synchronous from the caller's view (the lkh read-back is the sync point). */
int vbnmf_engine_step(vbnmf_engine *e, double aw, double bw, double ah, double bh, double fudge,
                      double *lkh, double *stats);

/* Split form of step for cell-partitioned engines.  step_local enqueues the local work
 * and leaves this partition's contribution in the reduce buffer (device memory, `count`
 * doubles); the caller all-reduces (sum) that buffer across partitions on the engine's
 * stream (RCCL), then step_finish produces lkh/stats (identical on every partition).
 * For an unpartitioned engine step == step_local; step_finish with no all-reduce. */
int vbnmf_engine_step_local(vbnmf_engine *e, double aw, double bw, double ah, double bh,
                            double fudge);
int vbnmf_engine_reduce_buffer(vbnmf_engine *e, void **device_ptr, int64_t *count);
int vbnmf_engine_step_finish(vbnmf_engine *e, double *lkh, double *stats);
/* After set_state on a partitioned engine the initial statistics need the same exchange:
 * set_state leaves them in the reduce buffer; all-reduce it, then call this. */
int vbnmf_engine_state_finish(vbnmf_engine *e);

/* The per-rank loop of vb_iterate (R/bayesian.R:336-352) run by the device: up to max_it steps with
 *   - hyper_update (R/bayesian.R:2-53, Niter = 100, Tol = 1e-3) after every step `it` with it > n0 and
 *     it %% dn == 0, `flags` = hyper.update (4 logicals),
 *   - break when lkh is NaN (:345), or when it > 1, it > n0, lkh >= lk0 and |1 - lkh/lk0| < tol (:346-347;
 *     lk0 then keeps the PREVIOUS step's evidence, as the reference leaves it), else lk0 <- lkh (:348).
 * Steps are queued ahead of the device, so no host round trip sits between two steps; kernels queued
 * beyond the break return at once and the state stays exactly as the breaking step left it.
 * hyper[4] = aw, bw, ah, bh (in: initial, out: final).  Outputs (any may be NULL): it = steps done,
 * lk0, lkh of the last step, reason (1 NaN, 2 converged, 3 hyper Newton did not converge -- the
 * reference stops with an error there, :43 -- 4 max_it reached), history[it][9] = lkh, mean log lw,
 * mean log lh, mean ew, mean eh, then aw, bw, ah, bh after that step's update (history_rows >= max_it).
 * A partitioned engine needs an RCCL communicator attached (below); every process then calls this with the
 * same arguments and gets the same results. */
int vbnmf_engine_run(vbnmf_engine *e, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0,
                     int32_t dn, const int32_t *flags, int32_t *it, double *lk0, double *lkh, int32_t *reason,
                     double *history, int64_t history_rows);

/* The restarts of one rank, stepped together.  Reference: vb_factorize runs `nrun` independent factorisations of every
 * rank, `lapply(seq_len(nrun), FUN = vb_iterate, bundle)` (R/bayesian.R:260-261; over Rmpi slaves at :262-263), each the loop of
 * R/bayesian.R:337-352 from its own random start.  On the small matrices ccfindR ships (inst/extdata: 1030 x 450) one such
 * loop cannot fill the GPU -- a step is two dependent launches of ~15 us for microseconds of work on a dozen workgroups -- and
 * concurrent streams or processes do not overlap them (profiles/r05_small_concurrent.txt), so the independent loops share
 * their launches instead: `count` engines of ONE rank on ONE matrix handle (same layouts, grids, update table; each engine
 * its own state, set beforehand), every step two launches for the whole batch, every engine following its own control block
 * (hyper-parameters, evidence, stop -- an engine that has stopped idles through the others' remaining steps).  Per engine the
 * results are those of vbnmf_engine_run on it alone, bit for bit.  hyper: [count][4] in / out; it_out, lk0_out, lkh_out,
 * reason_out: [count] (any may be NULL); history (or NULL): [count][history_rows][9], history_rows >= max_it.
 * Engines: unpartitioned, no communicator, padded rank <= 16, count <= 64, of ONE row width (one rank, or several ranks made
 * under vbnmf_set_engine_padding); the launches go on the first engine's stream. */
/* Grids of the engines the CALLING host thread creates from now on (0, 0: back to the defaults, one workgroup / block per
 * CU): engines meant for a batch of B want 256 / B of each -- B engines step in one launch of B x grid workgroups.  The
 * grid is part of the order of the block-wise sums: engines of different grids agree to rounding, not bit for bit.  (No
 * reference counterpart: launch geometry.) */
int vbnmf_set_engine_grid(int32_t n_wg, int32_t update_blocks);
/* Row width of the engines the CALLING host thread creates from now on: `padded_rank` columns (a padded rank >= the engine's
 * own: even up to 32, a multiple of 8 up to 64, of 16 beyond; 0: back to the rank's own), the columns beyond the rank held at
 * zero.  Engines of DIFFERENT ranks made this way share kernels, layouts and update table and may form one batch: the rank
 * loop of vb_iterate (R/bayesian.R:316) as well as the restarts step in one launch on a small matrix.  A padded engine agrees
 * with the unpadded one to rounding (the width fixes the update's thread mapping), not bit for bit. */
int vbnmf_set_engine_padding(int32_t padded_rank);
int vbnmf_batch_run(vbnmf_engine **engines, int32_t count, double *hyper, double fudge, int32_t max_it, double tol,
                    int32_t n0, int32_t dn, const int32_t *flags, int32_t *it_out, double *lk0_out, double *lkh_out,
                    int32_t *reason_out, double *history, int64_t history_rows);

/* ---------------------------------------------------------------------------------
 * Communicators for cell-partitioned runs.  The reference has no counterpart inside
 * an iteration: its only inter-process mechanism is Rmpi::mpi.applyLB over restarts (R/bayesian.R:262-263),
 * and vb_iterate's loop (R/bayesian.R:337-352) runs in one process.  Here one factorisation can span the GPUs
 * of a node, cells partitioned, one process per GPU, and the per-step exchange -- the sum over partitions of
 * [sw (n x r) | rowSums(eh) (r) | 4 scalars] -- is an RCCL all-reduce over xGMI enqueued by the LIBRARY on
 * the engine's streams, so a binding needs no collective of its own:
 *
 *   rank 0:   vbnmf_comm_unique_id(id)           then broadcast `id` (VBNMF_COMM_ID_BYTES) to every process
 *                                                 by whatever the host language has (Rmpi::mpi.bcast, MPI_Bcast,
 *                                                 torch.distributed.broadcast ...)
 *   all:      vbnmf_comm_create(id, ..., nranks, rank, device, &comm)        ncclCommInitRank
 *             vbnmf_engine_create_part(X, cb, ce, m, r, device, &e) ; vbnmf_engine_attach_comm(e, comm)
 *             vbnmf_engine_set_state(e, ...) ; vbnmf_engine_allreduce(e) ; vbnmf_engine_state_finish(e)
 *             vbnmf_engine_run(e, ...)           the whole loop of vb_iterate, device-driven, on every process;
 *                                                 per step the n x r piece of the all-reduce travels beside the
 *                                                 cell-side half of the sweep and a second, small one (the
 *                                                 sweeps' evidence partials, 8 KB) follows that sweep; every
 *                                                 process takes the same (replicated) stop decision and queues
 *                                                 the same collectives.  Every rank must run the same build
 *                                                 with the same VBNMF_NO_CONTROL_FOLD setting (it decides what
 *                                                 the second exchange carries).
 *        or   vbnmf_engine_step_local(e, ...) ; vbnmf_engine_allreduce(e) ; vbnmf_engine_step_finish(e, ...)
 *
 * librccl is opened at run time (dlopen "librccl.so.1"); without it vbnmf_comm_create fails with
 * VBNMF_ERR_NO_DEVICE and everything else keeps working.
 *
 * A LOCAL GROUP is the same protocol for partition engines that share ONE process and ONE device (RCCL
 * refuses two ranks on a device): tests and single-GPU rehearsals of a partitioned run.  The engines are
 * attached in partition order and driven together through vbnmf_group_state_finish / vbnmf_group_run; the
 * all-reduce is a kernel that adds the partitions' buffers in partition order.
 * --------------------------------------------------------------------------------- */
typedef struct vbnmf_comm vbnmf_comm;
#define VBNMF_COMM_ID_BYTES 128
int vbnmf_comm_unique_id(void *id, int64_t bytes);
int vbnmf_comm_create(const void *id, int64_t bytes, int32_t nranks, int32_t rank, int32_t device,
                      vbnmf_comm **out);
int vbnmf_comm_create_local(int32_t nranks, int32_t device, vbnmf_comm **out);
/* nranks, this process's rank (0 for a local group), kind (0 RCCL, 1 local group); any pointer may be NULL. */
int vbnmf_comm_info(const vbnmf_comm *c, int32_t *nranks, int32_t *rank, int32_t *kind);
void vbnmf_comm_destroy(vbnmf_comm *c);
/* The engine must live on the communicator's device.  An RCCL communicator takes one engine; a local group
 * takes its nranks partition engines, in partition order. */
int vbnmf_engine_attach_comm(vbnmf_engine *e, vbnmf_comm *c);
/* In-place all-reduce (sum, fp64) of the engine's reduce buffer on the engine's stream (RCCL communicators). */
int vbnmf_engine_allreduce(vbnmf_engine *e);
/* Local groups: the state exchange after set_state on every member, and the device-driven loop of the whole
 * group (arguments and results as vbnmf_engine_run; history comes from partition 0 -- all are identical). */
int vbnmf_group_state_finish(vbnmf_comm *c);
int vbnmf_group_run(vbnmf_comm *c, double *hyper, double fudge, int32_t max_it, double tol, int32_t n0,
                    int32_t dn, const int32_t *flags, int32_t *it, double *lk0, double *lkh, int32_t *reason,
                    double *history, int64_t history_rows);

/* Download the current wh members (any pointer may be NULL):
 * lw, ew, dw : n x r ; lh, eh, dh : r x m_local.  dw, dh are variances, as the reference
 * returns them (src/vbnmf_update.cpp:46,56); the R driver takes sqrt later (R/bayesian.R:382-383). */
int vbnmf_engine_get_state(vbnmf_engine *e, double *lw, double *lh, double *ew, double *eh,
                           double *dw, double *dh);

/* The HIP stream (hipStream_t) the engine launches on, for callers that order other
 * device work (an RCCL all-reduce) against it; and the option to adopt a caller's stream. */
int vbnmf_engine_get_stream(vbnmf_engine *e, void **stream);
int vbnmf_engine_set_stream(vbnmf_engine *e, void *stream);

/* Profiling aid for bench.py: when enabled, every step brackets the sweep kernel with HIP
 * events on the engine's stream; get returns the accumulated kernel time and launch
 * count since the last reset and resets them. */
int vbnmf_engine_timing_enable(vbnmf_engine *e, int32_t on);
int vbnmf_engine_timing_get(vbnmf_engine *e, double *sweep_ms, int64_t *sweep_launches);

/* Diagnostic (engine created with env VBNMF_DEBUG_TIMES=1): 100 MHz device timestamps of the last
 * sweep, out[side][wg][2 + 2*waves] = {workgroup start, end, then (start, end) per wave}. */
int vbnmf_engine_debug_times(vbnmf_engine *e, unsigned long long *out, int64_t capacity,
                             int32_t *n_wg, int32_t *waves);

/* Layout facts for roofline accounting / tests (any pointer may be NULL):
 * padded entry slots and bytes the two sweeps stream per step, task counts. */
int vbnmf_engine_layout_info(const vbnmf_engine *e, int64_t *nnz, int64_t *slots_gene_side,
                             int64_t *slots_cell_side, int64_t *stream_bytes_per_step,
                             int64_t *tasks_gene_side, int64_t *tasks_cell_side);

/* ---------------------------------------------------------------------------------
 * Stateless form: the reference's call, one X in, one updated `wh` out
 * (src/vbnmf_update.cpp:16-101).  Builds a throw-away engine on device 0 (or the device the environment variable VBNMF_DEVICE names); use the
 * engine API in a loop.  Outputs as the returned list's lw, lh, ew (= w), eh (= h),
 * dw, dh, lkh (:92-100).
 * --------------------------------------------------------------------------------- */
int vbnmf_update_dense(int64_t n, int64_t m, int32_t r, const double *X,
                       const double *lw_in, const double *lh_in, const double *eh_in,
                       double aw, double bw, double ah, double bh, double fudge,
                       double *lw, double *lh, double *ew, double *eh,
                       double *dw, double *dh, double *lkh);
/* The stateless entries (these two and vbnmf_ml_update_*) keep the LAST ingested matrix and its engine alive between
 * calls, keyed by the content of X: the dimensions, TWO independently seeded 64-bit hashes of every byte handed in (the
 * seed starts every chunk digest, so contents colliding under one seed do not under the other: a 128-bit key) and, on a
 * hit, a comparison of the CSC pointer array resp. of a strided sample (<= 4096 values) of the dense X.  The reference's
 * loop passes the same X thousands of times (R/bayesian.R:339), and a repeat then costs the hashes, the state transfer
 * and one step instead of ingestion + layouts + engine.  Results do not depend on it.  VBNMF_STATELESS_CACHE=0 disables it;
 * vbnmf_stateless_cache_clear() releases what is held (device and host memory). */
void vbnmf_stateless_cache_clear(void);
/* Device buffers of destroyed engines are kept in a per-process pool (at most VBNMF_POOL_MB, default 4096) for the next
 * engine: a rank sweep creates and destroys one engine per (run, rank) unit and every hipFree synchronises the device.
 * This returns what the pool holds to the driver. */
void vbnmf_pool_trim(void);
/* Same with X as dgCMatrix slots (no densification on the R side). */
int vbnmf_update_csc(int64_t n, int64_t m, int32_t r, const int32_t *p, const int32_t *i,
                     const double *x,
                     const double *lw_in, const double *lh_in, const double *eh_in,
                     double aw, double bw, double ah, double bh, double fudge,
                     double *lw, double *lh, double *ew, double *eh,
                     double *dw, double *dh, double *lkh);

/* ---------------------------------------------------------------------------------
 * Maximum-likelihood NMF on the same engine: the step of
 * factorize(), reference R/factorize.R:2-27 (nmf_updateR) with its likelihood :40-49
 * (the two are always called together, :195-196).  The factors live on the device
 * between steps; a VB state and an ML state exclude each other (setting one drops the
 * other).  Unpartitioned engines only.
 *
 *   ml_set_state  w : n x r column-major, h : r x m column-major (init, :30-38)
 *   ml_step       h <- h .* (t(w) %*% (x/(w h))) / colSums(w), clipped at double eps (:8-15);
 *                 then w <- w .* ((x/(w h)) %*% t(h)) / rowSums(h) on the NEW h (:17-24);
 *                 prior != 0 adds the Gamma prior terms (up + gamma_a - 1, down +
 *                 gamma_a/gamma_b, :10-13,19-22; factorize() itself never sets it).
 *                 *lk = likelihood(mat, w, h) of the updated pair (:40-49).
 *   ml_likelihood likelihood(mat, w, h) (:40-49) of the pair the engine holds now (after
 *                 ml_set_state: of the loaded pair; after ml_step: the value ml_step returned).
 *   ml_get_state  the current w, h (either may be NULL).
 * --------------------------------------------------------------------------------- */
int vbnmf_engine_ml_set_state(vbnmf_engine *e, const double *w, const double *h);
int vbnmf_engine_ml_step(vbnmf_engine *e, int32_t prior, double gamma_a, double gamma_b, double *lk);
/* factorize()'s inner loop under criterion = 'likelihood' (R/factorize.R:194-213) run by the device: up to
 * max_it steps, break when abs(lkold - lk) < tol * abs(lkold) (lkold starts at -Inf), steps queued ahead of
 * the GPU.  Outputs (any may be NULL): it = steps done, lk = likelihood of the last step, reason (2 converged,
 * 4 max_it reached), history[it] = the likelihood after every step (history_rows >= max_it). */
int vbnmf_engine_ml_run(vbnmf_engine *e, int32_t prior, double gamma_a, double gamma_b, int32_t max_it,
                        double tol, int32_t *it, double *lk, int32_t *reason, double *history,
                        int64_t history_rows);

/* The `nrun` restarts factorize() makes of every rank (`for(irun in seq_len(nrun))`, R/factorize.R:181; nrun defaults to 20),
 * their device-driven loops (vbnmf_engine_ml_run: R/factorize.R:194-213 under criterion = 'likelihood') stepped together: four
 * launches per step for the whole batch, every engine following its own control block.  Per engine the results are those of
 * vbnmf_engine_ml_run on it alone, bit for bit.  Engines as for vbnmf_batch_run (one rank, one matrix handle, rank <= 16,
 * count <= 64, grids from vbnmf_set_engine_grid), their states set by vbnmf_engine_ml_set_state.  it_out, lk_out, reason_out:
 * [count] (any may be NULL); history (or NULL): [count][history_rows], history_rows >= max_it. */
int vbnmf_batch_ml_run(vbnmf_engine **engines, int32_t count, int32_t prior, double gamma_a, double gamma_b, int32_t max_it,
                       double tol, int32_t *it_out, double *lk_out, int32_t *reason_out, double *history, int64_t history_rows);
int vbnmf_engine_ml_likelihood(vbnmf_engine *e, double *lk);
int vbnmf_engine_ml_get_state(vbnmf_engine *e, double *w, double *h);
/* Stateless forms of the same step: nmf_updateR(x, w, h, n, m, r, prior, gamma.a, gamma.b)
 * followed by likelihood(x, w, h) (R/factorize.R:2-27, :40-49); throw-away engine on device 0 (VBNMF_DEVICE overrides). */
int vbnmf_ml_update_dense(int64_t n, int64_t m, int32_t r, const double *X,
                          const double *w_in, const double *h_in,
                          int32_t prior, double gamma_a, double gamma_b,
                          double *w, double *h, double *lk);
int vbnmf_ml_update_csc(int64_t n, int64_t m, int32_t r, const int32_t *p, const int32_t *i,
                        const double *x, const double *w_in, const double *h_in,
                        int32_t prior, double gamma_a, double gamma_b,
                        double *w, double *h, double *lk);

/* Cluster of every cell from the coefficients the engine holds:
 * ids[j] = which.max(h[, j])[1], 1-based, first maximum on ties (R/factorize.R:55-56 inside
 * connectivity(), R/utils.R:906 cluster_id); h = the ML coefficient matrix, or E[H] of a VB
 * state.  ids: m_local int32.  Saves the r x m download when only the labels are needed
 * (the connectivity stopping criterion of factorize(), R/factorize.R:198-208). */
int vbnmf_engine_cluster_ids(vbnmf_engine *e, int32_t *ids);

/* The same labels compared with those of the PREVIOUS call: changed = sum(cnn != cnn0) over all pairs of cells, the
 * stopping count of factorize() under criterion = 'connectivity' (R/factorize.R:198-208; cnn = connectivity(h),
 * :51-60), formed on the device from the contingency table of the two labelings -- never from the O(m^2) pair
 * vectors.  changed = -1 on the first call after a state was loaded (the reference starts from npair, :200).
 * ids may be NULL. */
int vbnmf_engine_cluster_changes(vbnmf_engine *e, int64_t *changed, int32_t *ids);

/* vb_init(initializer = 'random') on the device (R/bayesian.R:111-115, 162-170): lw = ew ~ Gamma(shape aw, scale
 * bw/aw), lh = eh ~ Gamma(shape ah, scale bh/ah), dw = dh = 0, followed by what set_state does (a partitioned engine
 * needs the state exchange + state_finish).  Philox4x32-10 counters keyed by `seed`, Marsaglia-Tsang rejection; a
 * draw depends on (seed, factor, global element index) only, so partitions of one matrix draw pieces of the same H.
 * R's own RNG stream cannot be reproduced outside R: "identical seeds" means identical arrays, not R's numbers. */
int vbnmf_engine_random_state(vbnmf_engine *e, double aw, double bw, double ah, double bh, uint64_t seed);

/* Truncated SVD of the resident X, entirely on the device: what irlba::irlba(mat, rank) computes for the svd2
 * initialiser (R/bayesian.R:150-159).  Block subspace iteration on k = the engine's rank columns (create the engine
 * with rank = rank_out + oversampling): sparse products on the tiled layout, CholeskyQR2 orthonormalisation and a
 * k x k Jacobi eigen-solve per iteration; stops when the leading rank_out singular values move by <= tol * s[0], or
 * after maxit iterations.  u: n x rank_out, d: rank_out, vt: rank_out x m (column-major).  Drops any VB / ML state.
 * Fails with VBNMF_ERR_STATE if X has fewer than k independent directions.  Unpartitioned engines only. */
int vbnmf_engine_svd(vbnmf_engine *e, int32_t rank_out, double tol, int32_t maxit, uint64_t seed,
                     double *u, double *d, double *vt, int32_t *iterations);

/* ---------------------------------------------------------------------------------
 * Sparse products with the resident X, the two matrix-vector
 * blocks of a truncated SVD -- what irlba::irlba(mat, rank) computes for the svd2
 * initialiser (R/bayesian.R:150-159) -- on the same tiled layout as the update sweeps:
 *   transpose == 0 :  C (n x r) = X %*% t(B),   B : r x m  (column-major, like h)
 *   transpose != 0 :  C (r x m) = t(B) %*% X,   B : n x r  (column-major, like w)
 * r is the engine's rank.  Uses the factor arrays as operand storage: any VB / ML state
 * the engine held is dropped.  Unpartitioned engines only.
 * --------------------------------------------------------------------------------- */
int vbnmf_engine_spmm(vbnmf_engine *e, int32_t transpose, const double *B, double *C);

/* ---------------------------------------------------------------------------------
 * Host-only inspection of the tiled device layout (no GPU needed): builds the layout
 * for one side at padded rank r and hands out its arrays so tests can check, bit for
 * bit, that the slices hold exactly X.  side 0 = gene side (lanes own genes, minor =
 * cells), 1 = cell side.  The returned pointers belong to the layout object.
 * --------------------------------------------------------------------------------- */
typedef struct vbnmf_layout vbnmf_layout;
typedef struct {
    int32_t side, wide;            /* wide: 0 = 4-byte entries (integer counts; one above 16383 takes several slots of the
                                      same minor), 1 = u32 index + f64 value (non-integer X) */
    int64_t n_major, n_minor;      /* lanes own majors; minors are gathered from LDS */
    int32_t block_width;           /* minors in the widest LDS block (blocks differ: see block_start) */
    int32_t n_blocks;              /* ceil(n_minor / block_width) */
    int32_t max_len;               /* longest task (entries per lane) */
    int32_t n_wg;                  /* persistent workgroups the work list is cut for */
    int32_t row_slots;             /* 16-byte LDS slots per staged factor row at this rank */
    int64_t n_tasks, n_slices;     /* task = run of one major's entries in one block; slice = 64 tasks */
    int64_t n_slots;               /* padded entry slots (all slices) */
    int64_t n_segs;                /* segment = the slices of one block in one workgroup's share */
    const uint32_t *task_major;    /* [n_slices*64] major of task slice*64+lane, 0xFFFFFFFF = idle lane */
    const int32_t *slice_width;    /* [n_slices] entries per lane (multiple of 4) */
    const int64_t *slice_off;      /* [n_slices] first slot of the slice; slot(t, lane) =
                                      off + (t/4)*256 + lane*4 + t%4 */
    const int32_t *slice_block;    /* [n_slices] minor block */
    const int32_t *slice_fast;     /* [n_slices] low 16 bits: leading entries per lane that are stored ones (value exactly 1) in
                                      EVERY lane of the slice (multiple of 8; 0 for wide layouts): the sweep runs them through
                                      a shorter loop; high 16 bits: leading entries that are ones or twos in every lane (>= the
                                      low half; the gene side defers their logarithm).  A task's ones come first, then its twos,
                                      then its other entries. */
    const int64_t *block_start;    /* [n_blocks+1] first minor of each block: boundaries sit where the cumulative entry
                                      count reaches a whole number of workgroup quotas */
    const int32_t *seg_block;      /* [n_segs] */
    const int32_t *wg_seg0;        /* [n_wg+1] segments of each workgroup */
    const int32_t *seg_ptr;        /* [n_segs+1] first slice of each segment; slices are numbered in processing order */
    const int32_t *inv_ptr;        /* [n_major+1] tasks of each major ... */
    const uint32_t *inv_task;      /* [n_tasks]   ... in the order their partials are summed */
    const uint32_t *packed;        /* [n_slots] (count << 18) | (local minor * row_slots << 4)   (wide == 0) */
    const uint32_t *wide_idx;      /* [n_slots] local minor                    (wide == 1) */
    const double *wide_val;        /* [n_slots]                                (wide == 1) */
    const int32_t *cell_perm;      /* [cells of the column range] or NULL: the layout's internal renumbering of the cells
                                      (minors on side 0, majors on side 1): position -> column relative to col_begin.
                                      NULL = as stored.  Cells with alike gene support are stored next to each other, so
                                      a gene's entries fall into fewer cell blocks (csrc/order.cpp; VBNMF_CELL_ORDER=0/1). */
} vbnmf_layout_view;

int vbnmf_layout_build(const vbnmf_matrix *X, int64_t col_begin, int64_t col_end, int32_t side,
                       int32_t r, vbnmf_layout **out, vbnmf_layout_view *view);
void vbnmf_layout_destroy(vbnmf_layout *L);

/* ---------------------------------------------------------------------------------
 * Test hooks: the library's own fp64 ln / digamma / lnGamma (which stand in for libm's log
 * and GSL's gsl_sf_psi / gsl_sf_lngamma, reference src/vbnmf_update.cpp:59,63,73,81-89),
 * evaluated on the host build and on the device, so tests can check them against mpmath.
 * kind: 0 = ln(x), 1 = psi(x), 2 = lnGamma(x), 3 = 1/x, 4/5 = raw hardware reciprocal seeds
 * (device only), 6 = the sweep's table-driven ln(x).
 * --------------------------------------------------------------------------------- */
int vbnmf_test_special_host(int32_t kind, int64_t n, const double *x, double *y);
int vbnmf_test_special_device(int32_t kind, int64_t n, const double *x, double *y);
/* Test hook for the bounded waits: enqueues a host function that sleeps `seconds` (<= 60) on the engine's stream, so
 * the steps queued behind it are late without the GPU being busy.  The waits of vbnmf_engine_step / _run / _ml_run are
 * bounded by VBNMF_WAIT_TIMEOUT_S (seconds of wall clock, default 300): past it the call returns VBNMF_ERR_HIP with a
 * message naming the last completed step, and the engine is left untouched (its destroy neither waits nor frees while
 * work is still queued).  The reference has no analogue: its call is synchronous CPU code (src/RcppExports.cpp:11-22). */
int vbnmf_test_stream_sleep(vbnmf_engine *e, double seconds);
/* Test hook (host only): the stateless cache's content hash of `bytes` bytes under `seed`. */
uint64_t vbnmf_test_hash_bytes(const void *data, int64_t bytes, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* VBNMF_H */
