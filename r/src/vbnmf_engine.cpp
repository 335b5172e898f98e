// r/src/vbnmf_engine.cpp -- the RESIDENT binding (INTEGRATION.md sections 2, 2b, 2b', 2c, 2d, 3): X and wh stay on the GPU for a
// whole (run, rank) factorisation.  Add this file to <ccfindR>/src beside r/src/vbnmf_update.cpp, re-run
// Rcpp::compileAttributes() (it regenerates src/RcppExports.cpp / R/RcppExports.R with these exports added; the drop-in symbol
// _ccfindR_vbnmf_update keeps its name and arity) and source r/R/hip_backend.R for the R side of the loop.
//
// Reference code these exports stand in for, by export:
//   vbnmf_matrix / vbnmf_matrix_mtx   counts(object) handed to every vbnmf_update call (R/bayesian.R:239, :339);
//                                     read_10x's as(Matrix::readMM(count), 'dgCMatrix') (R/utils.R:34)
//   vbnmf_engine(_geom) / vbnmf_step  one iteration of vb_iterate's loop: wh <- vbnmf_update(...) (R/bayesian.R:339)
//   vbnmf_state                       the wh members vb_iterate reads after the loop (R/bayesian.R:379-383)
//   vbnmf_run                         the whole loop (R/bayesian.R:337-352), hyper_update (:2-53) included
//   vbnmf_run_batch, vbnmf_set_grid   the nrun restarts of a rank (R/bayesian.R:260-261), their loops stepped by one launch
//   vbnmf_rank_classes                the rank loop of vb_iterate (R/bayesian.R:316) sharing one pair of layouts
//   mlnmf_*                           nmf_updateR + likelihood (R/factorize.R:2-27, :40-49; called at :195-196)
//   vbnmf_comm*, vbnmf_engine_part    no counterpart: the reference's only inter-process mechanism is Rmpi::mpi.applyLB over
//                                     restarts (R/bayesian.R:262-263)
// Every vbnmf_* function called here is declared in include/vbnmf.h (tests/test_r_binding_symbols.py checks names and arity).
#include <Rcpp.h>
#include "vbnmf.h"

static void check(int rc) { if (rc != VBNMF_OK) Rcpp::stop("vbnmf: %s", vbnmf_last_error()); }
static void engine_finalizer(SEXP p) { vbnmf_engine_destroy((vbnmf_engine *)R_ExternalPtrAddr(p)); R_ClearExternalPtr(p); }
static void matrix_finalizer(SEXP p) { vbnmf_matrix_destroy((vbnmf_matrix *)R_ExternalPtrAddr(p)); R_ClearExternalPtr(p); }
static void comm_finalizer(SEXP p) { vbnmf_comm_destroy((vbnmf_comm *)R_ExternalPtrAddr(p)); R_ClearExternalPtr(p); }

static SEXP wrap_matrix(vbnmf_matrix *M)
{
    SEXP h = PROTECT(R_MakeExternalPtr(M, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(h, matrix_finalizer, TRUE);
    UNPROTECT(1);
    return h;
}
static SEXP wrap_engine(vbnmf_engine *e, SEXP matrix)
{
    SEXP h = PROTECT(R_MakeExternalPtr(e, R_NilValue, matrix));            // keeps the matrix alive
    R_RegisterCFinalizerEx(h, engine_finalizer, TRUE);
    UNPROTECT(1);
    return h;
}
static vbnmf_engine *eng(SEXP h)
{
    vbnmf_engine *e = (vbnmf_engine *)R_ExternalPtrAddr(h);
    if (!e) Rcpp::stop("vbnmf: the engine handle is stale");
    return e;
}

// [[Rcpp::export]]
SEXP vbnmf_matrix(SEXP mat)                                  // dense matrix or dgCMatrix -- ingested ONCE per vb_factorize
{
    vbnmf_matrix *M = nullptr;
    if (Rf_isS4(mat)) {                                      // Matrix::dgCMatrix slots @Dim, @p, @i, @x (R/utils.R:34)
        Rcpp::S4 s(mat);
        Rcpp::IntegerVector dim = s.slot("Dim"), p = s.slot("p"), i = s.slot("i");
        Rcpp::NumericVector x = s.slot("x");
        check(vbnmf_matrix_from_csc(dim[0], dim[1], p.begin(), i.begin(), x.begin(), &M));
    } else {
        Rcpp::NumericMatrix X(mat);
        check(vbnmf_matrix_from_dense(X.nrow(), X.ncol(), X.begin(), &M));
    }
    return wrap_matrix(M);
}

// [[Rcpp::export]]
SEXP vbnmf_matrix_mtx(std::string path)                      // native parallel Matrix Market reader (read_10x, R/utils.R:34)
{
    vbnmf_matrix *M = nullptr;
    check(vbnmf_matrix_from_mtx(path.c_str(), &M));
    return wrap_matrix(M);
}

// [[Rcpp::export]]
Rcpp::IntegerVector vbnmf_rank_classes(Rcpp::IntegerVector ranks, int max_classes = 1)
{
    Rcpp::IntegerVector out(std::max<R_xlen_t>(ranks.size(), 1));
    int32_t n = 0;
    check(vbnmf_plan_classes(ranks.begin(), (int32_t)ranks.size(), max_classes, out.begin(), &n));
    return Rcpp::head(out, n);                               // padded ranks, ascending; one class = the largest rank
}

// [[Rcpp::export]]
int vbnmf_padded(int rank) { return vbnmf_padded_rank(rank); }

// geometry_rank: 0 = the rank's own geometry, else the sweep's class (vbnmf_rank_classes); wh: list(lw, lh, eh) or NULL
// [[Rcpp::export]]
SEXP vbnmf_engine_geom(SEXP matrix, int rank_k, int geometry_rank, SEXP wh, int device = 0)
{
    vbnmf_matrix *M = (vbnmf_matrix *)R_ExternalPtrAddr(matrix);
    int64_t m = 0;
    check(vbnmf_matrix_info(M, nullptr, &m, nullptr, nullptr));
    vbnmf_engine *e = nullptr;
    check(vbnmf_engine_create_geom(M, 0, m, m, rank_k, geometry_rank, device, &e));
    if (!Rf_isNull(wh)) {
        Rcpp::List w(wh);
        Rcpp::NumericMatrix lw = w["lw"], lh = w["lh"], eh = w["eh"];
        const int rc = vbnmf_engine_set_state(e, lw.begin(), lh.begin(), eh.begin());
        if (rc) { vbnmf_engine_destroy(e); check(rc); }
    }
    return wrap_engine(e, matrix);
}

// [[Rcpp::export]]
SEXP vbnmf_engine(SEXP matrix, int rank_k, SEXP wh, int device = 0) { return vbnmf_engine_geom(matrix, rank_k, 0, wh, device); }

// [[Rcpp::export]]
void vbnmf_set_state(SEXP engine, const Rcpp::List &wh)
{
    Rcpp::NumericMatrix lw = wh["lw"], lh = wh["lh"], eh = wh["eh"];
    check(vbnmf_engine_set_state(eng(engine), lw.begin(), lh.begin(), eh.begin()));
}

// one iteration: replaces wh <- vbnmf_update(as.matrix(bundle$mat), wh, hyper, c(bundle$fudge))  (R/bayesian.R:339)
// [[Rcpp::export]]
Rcpp::List vbnmf_step(SEXP engine, const Rcpp::List &hyper, double fudge)
{
    double lkh, st[4];
    check(vbnmf_engine_step(eng(engine), hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"], fudge, &lkh, st));
    return Rcpp::List::create(Rcpp::Named("lkh") = lkh, Rcpp::Named("lwm") = st[0], Rcpp::Named("lhm") = st[1],
                              Rcpp::Named("ewm") = st[2], Rcpp::Named("ehm") = st[3]);
}

// the wh members of the reference's return list (src/vbnmf_update.cpp:92-100), downloaded once after the loop
// [[Rcpp::export]]
Rcpp::List vbnmf_state(SEXP engine)
{
    int64_t n = 0, m = 0;
    int32_t r = 0;
    check(vbnmf_engine_dims(eng(engine), &n, &m, &r));
    Rcpp::NumericMatrix lw(n, r), ew(n, r), dw(n, r), lh(r, m), eh(r, m), dh(r, m);
    check(vbnmf_engine_get_state(eng(engine), lw.begin(), lh.begin(), ew.begin(), eh.begin(), dw.begin(), dh.begin()));
    return Rcpp::List::create(Rcpp::Named("w") = ew, Rcpp::Named("h") = eh, Rcpp::Named("lw") = lw, Rcpp::Named("lh") = lh,
                              Rcpp::Named("ew") = ew, Rcpp::Named("eh") = eh, Rcpp::Named("dw") = dw, Rcpp::Named("dh") = dh);
}

// the whole per-rank loop (R/bayesian.R:337-352) on the device
// [[Rcpp::export]]
Rcpp::List vbnmf_run(SEXP engine, Rcpp::NumericVector hyper, double fudge, int Itmax, double Tol, int n0, int dn,
                     Rcpp::LogicalVector hyper_update)
{
    double hy[4] = {hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"]}, lk0 = 0.0, lkh = 0.0;
    int32_t fl[4] = {hyper_update[0], hyper_update[1], hyper_update[2], hyper_update[3]}, it = 0, reason = 0;
    check(vbnmf_engine_run(eng(engine), hy, fudge, Itmax, Tol, n0, dn, fl, &it, &lk0, &lkh, &reason, nullptr, 0));
    if (reason == 3) Rcpp::stop("Hyperparameter update failed to converge");            // R/bayesian.R:43
    return Rcpp::List::create(Rcpp::Named("it") = it, Rcpp::Named("lk0") = lk0, Rcpp::Named("lkh") = lkh,
                              Rcpp::Named("reason") = reason,
                              Rcpp::Named("hyper") = Rcpp::NumericVector::create(Rcpp::Named("aw") = hy[0],
                                  Rcpp::Named("bw") = hy[1], Rcpp::Named("ah") = hy[2], Rcpp::Named("bh") = hy[3]));
}

// the restarts of ONE rank stepped together (R/bayesian.R:260-261: lapply(seq_len(nrun), vb_iterate), here rank by rank): one
// launch steps every engine of the list -- on the small matrices ccfindR ships a single loop cannot fill the GPU.  `engines`
// were made (with their states) after vbnmf_set_grid(256 %/% B, 256 %/% B); hyper: B x 4 matrix (aw, bw, ah, bh per engine).
// [[Rcpp::export]]
void vbnmf_set_grid(int n_wg = 0, int update_blocks = 0) { check(vbnmf_set_engine_grid(n_wg, update_blocks)); }

// engines of SEVERAL ranks in one batch (the rank loop, R/bayesian.R:316): made after vbnmf_set_padding(vbnmf_padded(max(ranks)))
// they are all as wide as the widest rank's; vbnmf_set_padding(0) afterwards
// [[Rcpp::export]]
void vbnmf_set_padding(int padded_rank = 0) { check(vbnmf_set_engine_padding(padded_rank)); }

// [[Rcpp::export]]
Rcpp::List vbnmf_run_batch(Rcpp::List engines, Rcpp::NumericMatrix hyper, double fudge, int Itmax, double Tol, int n0, int dn,
                           Rcpp::LogicalVector hyper_update)
{
    const int B = engines.size();
    if (hyper.nrow() != B || hyper.ncol() != 4) Rcpp::stop("vbnmf: hyper must be a B x 4 matrix");
    std::vector<vbnmf_engine *> es(B);
    std::vector<double> hy((size_t)B * 4), lk0(B), lkh(B);
    std::vector<int32_t> it(B), reason(B);
    for (int b = 0; b < B; b++) { es[b] = eng(engines[b]); for (int q = 0; q < 4; q++) hy[(size_t)b * 4 + q] = hyper(b, q); }
    int32_t fl[4] = {hyper_update[0], hyper_update[1], hyper_update[2], hyper_update[3]};
    check(vbnmf_batch_run(es.data(), B, hy.data(), fudge, Itmax, Tol, n0, dn, fl, it.data(), lk0.data(), lkh.data(), reason.data(), nullptr, 0));
    Rcpp::NumericMatrix out(B, 4);
    Rcpp::IntegerVector its(B), why(B);
    Rcpp::NumericVector l0(B), lh(B);
    for (int b = 0; b < B; b++) {
        if (reason[b] == 3) Rcpp::stop("Hyperparameter update failed to converge");      // R/bayesian.R:43
        for (int q = 0; q < 4; q++) out(b, q) = hy[(size_t)b * 4 + q];
        its[b] = it[b]; why[b] = reason[b]; l0[b] = lk0[b]; lh[b] = lkh[b];
    }
    Rcpp::colnames(out) = Rcpp::CharacterVector::create("aw", "bw", "ah", "bh");
    return Rcpp::List::create(Rcpp::Named("it") = its, Rcpp::Named("lk0") = l0, Rcpp::Named("lkh") = lh,
                              Rcpp::Named("reason") = why, Rcpp::Named("hyper") = out);
}

// ---- factorize(): nmf_updateR + likelihood (R/factorize.R:2-27, :40-49)
// [[Rcpp::export]]
void mlnmf_set_state(SEXP engine, const Rcpp::NumericMatrix &w, const Rcpp::NumericMatrix &h)
{
    check(vbnmf_engine_ml_set_state(eng(engine), w.begin(), h.begin()));
}
// [[Rcpp::export]]
double mlnmf_step(SEXP engine, bool prior = false, double gamma_a = 1.0, double gamma_b = 1.0)
{
    double lk;
    check(vbnmf_engine_ml_step(eng(engine), prior ? 1 : 0, gamma_a, gamma_b, &lk));
    return lk;
}
// [[Rcpp::export]]
Rcpp::List mlnmf_run(SEXP engine, int Itmax, double Tol, bool prior = false, double gamma_a = 1.0, double gamma_b = 1.0)
{
    int32_t it = 0, reason = 0;
    double lk = 0.0;
    check(vbnmf_engine_ml_run(eng(engine), prior ? 1 : 0, gamma_a, gamma_b, Itmax, Tol, &it, &lk, &reason, nullptr, 0));
    return Rcpp::List::create(Rcpp::Named("it") = it, Rcpp::Named("lk") = lk, Rcpp::Named("reason") = reason);
}

// the nrun restarts of a rank (R/factorize.R:181) stepped together: engines made after vbnmf_set_grid(256 %/% B, 256 %/% B),
// their starts loaded with mlnmf_set_state; one call for all their loops (R/factorize.R:194-213 each)
// [[Rcpp::export]]
Rcpp::List mlnmf_run_batch(Rcpp::List engines, int Itmax, double Tol, bool prior = false, double gamma_a = 1.0, double gamma_b = 1.0)
{
    const int B = engines.size();
    std::vector<vbnmf_engine *> es(B);
    std::vector<int32_t> it(B), reason(B);
    std::vector<double> lk(B);
    for (int b = 0; b < B; b++) es[b] = eng(engines[b]);
    check(vbnmf_batch_ml_run(es.data(), B, prior ? 1 : 0, gamma_a, gamma_b, Itmax, Tol, it.data(), lk.data(), reason.data(), nullptr, 0));
    return Rcpp::List::create(Rcpp::Named("it") = Rcpp::IntegerVector(it.begin(), it.end()),
                              Rcpp::Named("lk") = Rcpp::NumericVector(lk.begin(), lk.end()),
                              Rcpp::Named("reason") = Rcpp::IntegerVector(reason.begin(), reason.end()));
}
// [[Rcpp::export]]
Rcpp::List mlnmf_state(SEXP engine)
{
    int64_t n = 0, m = 0;
    int32_t r = 0;
    check(vbnmf_engine_dims(eng(engine), &n, &m, &r));
    Rcpp::NumericMatrix w(n, r), h(r, m);
    check(vbnmf_engine_ml_get_state(eng(engine), w.begin(), h.begin()));
    return Rcpp::List::create(Rcpp::Named("ew") = w, Rcpp::Named("eh") = h);
}
// which.max per cell (R/factorize.R:55-56) without downloading h
// [[Rcpp::export]]
Rcpp::IntegerVector vbnmf_cluster_ids(SEXP engine)
{
    int64_t m = 0;
    check(vbnmf_engine_dims(eng(engine), nullptr, &m, nullptr));
    Rcpp::IntegerVector ids(m);
    check(vbnmf_engine_cluster_ids(eng(engine), ids.begin()));
    return ids;
}

// ---- one factorisation, cells partitioned over the GPUs: the exchange is the library's
// [[Rcpp::export]]
Rcpp::RawVector vbnmf_comm_id()                      // on rank 0; then Rmpi::mpi.bcast(id, type = 4 /* raw */, rank = 0)
{
    Rcpp::RawVector id(VBNMF_COMM_ID_BYTES);
    check(vbnmf_comm_unique_id(&id[0], id.size()));
    return id;
}
// [[Rcpp::export]]
SEXP vbnmf_comm(Rcpp::RawVector id, int nranks, int rank, int device)       // ncclCommInitRank: call on every process
{
    vbnmf_comm *c = nullptr;
    check(vbnmf_comm_create(&id[0], id.size(), nranks, rank, device, &c));
    SEXP h = PROTECT(R_MakeExternalPtr(c, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(h, comm_finalizer, TRUE);
    UNPROTECT(1);
    return h;
}
// [[Rcpp::export]]
SEXP vbnmf_engine_part(SEXP matrix, SEXP comm, int rank_k, double col_begin, double col_end, const Rcpp::List &wh, int device)
{
    vbnmf_matrix *M = (vbnmf_matrix *)R_ExternalPtrAddr(matrix);
    int64_t m = 0;
    check(vbnmf_matrix_info(M, nullptr, &m, nullptr, nullptr));
    vbnmf_engine *e = nullptr;
    check(vbnmf_engine_create_part(M, (int64_t)col_begin, (int64_t)col_end, m, rank_k, device, &e));
    SEXP h = wrap_engine(e, matrix);                                        // (owned from here on: errors below free it through the GC)
    PROTECT(h);
    check(vbnmf_engine_attach_comm(e, (vbnmf_comm *)R_ExternalPtrAddr(comm)));
    Rcpp::NumericMatrix lw = wh["lw"], lh = wh["lh"], eh = wh["eh"];       // lh, eh: this process's columns only
    check(vbnmf_engine_set_state(e, lw.begin(), lh.begin(), eh.begin()));
    check(vbnmf_engine_allreduce(e));                                       // the statistics of the loaded state
    check(vbnmf_engine_state_finish(e));
    UNPROTECT(1);
    return h;
}
