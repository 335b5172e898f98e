# r/R/hip_backend.R -- the R side of the resident binding (INTEGRATION.md section 2): what changes in the reference's driver
# when `backend = "hip"` is chosen next to `useC`.  Source it into a ccfindR checkout whose src/ holds r/src/*.cpp.
# Nothing here is needed for the plain drop-in (r/src/vbnmf_update.cpp alone): that one keeps R/bayesian.R untouched.

# hyper_update (reference R/bayesian.R:2-53) reads wh only through four means (:8-11: mean(log(wh$lw)), mean(log(wh$lh)),
# mean(wh$ew), mean(wh$eh)).  The engine returns exactly those four numbers with every step, so the reference's OWN function
# is called, unchanged, on a one-element stand-in for wh whose means are those numbers (log(exp(x)) == x to one ulp): its
# Newton recurrences, step halving, stopping rule, error and closing assignments are not re-stated here.
hyper_update_means <- function(hyper.update, s, hyper, Niter = 100, Tol = 1e-3)
  hyper_update(hyper.update, list(lw = exp(s$lwm), lh = exp(s$lhm), ew = s$ewm, eh = s$ehm), hyper, Niter = Niter, Tol = Tol)

# The per-rank loop of vb_iterate (reference R/bayesian.R:334-352) over the resident engine, one read-back per iteration;
# `gpu_mat` = vbnmf_matrix(mat), made once in vb_factorize.  device_loop = TRUE hands the whole loop to the device instead
# (vbnmf_run: one call per (run, rank)).
vb_iterate_rank_hip <- function(gpu_mat, rank, wh, hyper, bundle, geometry_rank = 0, device = 0, device_loop = TRUE) {
  eng <- vbnmf_engine_geom(gpu_mat, rank, geometry_rank, wh, device)
  if (device_loop) {
    out <- vbnmf_run(eng, unlist(hyper), bundle$fudge, bundle$Itmax, bundle$Tol, bundle$hyper.update.n0,
                     bundle$hyper.update.dn, bundle$hyper.update)
    return(list(wh = vbnmf_state(eng), hyper = as.list(out$hyper), lk0 = out$lk0, it = out$it))
  }
  lk0 <- 0
  for (it in seq_len(bundle$Itmax)) {
    s <- vbnmf_step(eng, hyper, bundle$fudge)                                      # replaces :339
    if (it > bundle$hyper.update.n0 & it %% bundle$hyper.update.dn == 0)
      hyper <- hyper_update_means(bundle$hyper.update, s, hyper, Niter = 100, Tol = 1e-3)   # :342-344
    if (is.na(s$lkh)) break                                                        # :345
    if (it > 1) if (it > bundle$hyper.update.n0)
      if (s$lkh >= lk0) if (abs(1 - s$lkh / lk0) < bundle$Tol) break               # :346-347
    lk0 <- s$lkh                                                                   # :348
  }
  list(wh = vbnmf_state(eng), hyper = hyper, lk0 = lk0, it = it)
}

# The nrun restarts of ONE rank (reference R/bayesian.R:260-261 runs them one after the other, rank loop inside each) stepped
# TOGETHER: one engine per restart, made on the grid a batch of B wants, then one call for all their loops (vbnmf_run_batch).
# `whs` = list of B initial states from vb_init (R/bayesian.R:331), `hyper` the common starting values (:321-326).  On the
# small matrices ccfindR ships (inst/extdata: 1030 x 450) this is where the GPU's parallelism is: 23 000 -> 160 000
# iterations per second in all for 32 restarts (profiles/r05_small_concurrent.txt).
vb_restarts_rank_hip <- function(gpu_mat, rank, whs, hyper, bundle, geometry_rank = 0, device = 0) {
  B <- length(whs)
  g <- max(8, (256 %/% B) %/% 8 * 8)
  vbnmf_set_grid(g, g)
  engs <- tryCatch(lapply(whs, function(wh) vbnmf_engine_geom(gpu_mat, rank, geometry_rank, wh, device)),
                   finally = vbnmf_set_grid(0, 0))
  hy <- matrix(rep(unlist(hyper[c("aw", "bw", "ah", "bh")]), each = B), nrow = B)
  out <- vbnmf_run_batch(engs, hy, bundle$fudge, bundle$Itmax, bundle$Tol, bundle$hyper.update.n0,
                         bundle$hyper.update.dn, bundle$hyper.update)
  lapply(seq_len(B), function(b) list(wh = vbnmf_state(engs[[b]]), hyper = as.list(out$hyper[b, ]), lk0 = out$lk0[b],
                                      it = out$it[b]))
}
