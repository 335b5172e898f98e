// r/src/vbnmf_update.cpp -- ccfindR's native step on the MI355X engine: the DROP-IN for the reference's
// src/vbnmf_update.cpp (variant (a) of INTEGRATION.md section 1: the reference's C++ signature is kept, so its generated glue
// src/RcppExports.cpp, R/RcppExports.R and NAMESPACE stay byte for byte as they are and only GSL leaves src/Makevars).
//
//   cp r/src/vbnmf_update.cpp  <ccfindR>/src/vbnmf_update.cpp
//   cp r/src/Makevars          <ccfindR>/src/Makevars            (set VBNMF_HOME)
//   R CMD INSTALL <ccfindR>
//
// Reference interface replaced: Rcpp::List vbnmf_update(const Eigen::MatrixXd&, const Rcpp::List&, const Rcpp::List&,
// const Rcpp::NumericVector&) (src/vbnmf_update.cpp:16-17), reached through .Call("_ccfindR_vbnmf_update", ...)
// (src/RcppExports.cpp:11-22, R/RcppExports.R:4-6), only caller vb_iterate (R/bayesian.R:339).
// C ABI called: vbnmf_update_dense, vbnmf_last_error (include/vbnmf.h).
//
// -DVBNMF_STANDALONE_SHIM builds the same function as a library of its own with the glue included (r/tests/parity.R loads it
// BESIDE an installed, unmodified ccfindR and calls both through .Call(..., PACKAGE = )).
#include <RcppEigen.h>
#include "vbnmf.h"

static void check(int rc) { if (rc != VBNMF_OK) Rcpp::stop("vbnmf: %s", vbnmf_last_error()); }

// [[Rcpp::depends(RcppEigen)]]
// [[Rcpp::export]]
Rcpp::List vbnmf_update(const Eigen::MatrixXd &X, const Rcpp::List &wh,
                        const Rcpp::List &hyper, const Rcpp::NumericVector &fudge)
{
    const double fud = fudge[0];                                   // src/vbnmf_update.cpp:19
    Rcpp::NumericMatrix lw0 = wh["lw"], lh0 = wh["lh"], eh0 = wh["eh"];   // :22-25 (ew is overwritten at :44)
    const int64_t n = X.rows(), m = X.cols();
    const int r = lw0.ncol();                                      // :27
    if (lw0.nrow() != n || lh0.nrow() != r || lh0.ncol() != m || eh0.nrow() != r || eh0.ncol() != m)
        Rcpp::stop("vbnmf_update: wh members do not match X");
    const double aw = hyper["aw"], ah = hyper["ah"], bw = hyper["bw"], bh = hyper["bh"];   // :28-31
    Rcpp::NumericMatrix lw(n, r), ew(n, r), dw(n, r), lh(r, m), eh(r, m), dh(r, m);
    double lkh = NA_REAL;
    check(vbnmf_update_dense(n, m, r, X.data(), lw0.begin(), lh0.begin(), eh0.begin(),
                             aw, bw, ah, bh, fud,
                             lw.begin(), lh.begin(), ew.begin(), eh.begin(), dw.begin(), dh.begin(), &lkh));
    return Rcpp::List::create(Rcpp::Named("w") = ew, Rcpp::Named("h") = eh, Rcpp::Named("lw") = lw,
                              Rcpp::Named("lh") = lh, Rcpp::Named("ew") = ew, Rcpp::Named("eh") = eh,
                              Rcpp::Named("lkh") = lkh, Rcpp::Named("dw") = dw, Rcpp::Named("dh") = dh);   // :92-100
}

#ifdef VBNMF_STANDALONE_SHIM
// The glue Rcpp::compileAttributes() generates for the function above (same symbol, arity 4, registration as the reference's
// src/RcppExports.cpp:11-32), under the library's own name so that it can be loaded beside an installed ccfindR.
RcppExport SEXP _ccfindR_vbnmf_update(SEXP XSEXP, SEXP whSEXP, SEXP hyperSEXP, SEXP fudgeSEXP)
{
BEGIN_RCPP
    Rcpp::RObject rcpp_result_gen;
    Rcpp::RNGScope rcpp_rngScope_gen;
    Rcpp::traits::input_parameter<const Eigen::MatrixXd &>::type X(XSEXP);
    Rcpp::traits::input_parameter<const Rcpp::List &>::type wh(whSEXP);
    Rcpp::traits::input_parameter<const Rcpp::List &>::type hyper(hyperSEXP);
    Rcpp::traits::input_parameter<const Rcpp::NumericVector &>::type fudge(fudgeSEXP);
    rcpp_result_gen = Rcpp::wrap(vbnmf_update(X, wh, hyper, fudge));
    return rcpp_result_gen;
END_RCPP
}
static const R_CallMethodDef CallEntries[] = {
    {"_ccfindR_vbnmf_update", (DL_FUNC)&_ccfindR_vbnmf_update, 4},
    {NULL, NULL, 0}
};
RcppExport void R_init_vbnmf_shim(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
#endif
