#!/usr/bin/env Rscript
# r/tests/parity.R -- PINS THE PARITY that this repository cannot pin by itself (it has no R): run it on a machine that has
# R and an installed, UNMODIFIED ccfindR.  For every committed single-step fixture (tests/golden/step_*.npz, exported by
# r/tests/export_goldens.py) and for the reference's bundled PBMC sample it prints the maximum relative error, output by
# output, between
#   (1) the reference's R twin        ccfindR:::vbnmf_updateR   (R/bayesian.R:56-106)
#   (2) the reference's native step   ccfindR:::vbnmf_update    (src/vbnmf_update.cpp:16-102 via R/RcppExports.R:4-6)
#   (3) the fixture's expected outputs (this repository's CPU oracle, oracle/vbnmf_oracle.c)
#   (4) the MI355X shim, if built:     R CMD SHLIB -o vbnmf_shim.so -DVBNMF_STANDALONE_SHIM r/src/vbnmf_update.cpp (with
#                                      r/src/Makevars' flags) on a machine with the GPU library; VBNMF_SHIM=/path/to/vbnmf_shim.so
# Tolerances (SURVEY.md section 8c): factors 1e-12, lkh 1e-10.  Exit status 1 when any comparison exceeds them.
#
#   python3 r/tests/export_goldens.py && Rscript r/tests/parity.R [r/tests/golden_bin]
suppressMessages(library(ccfindR))
args <- commandArgs(trailingOnly = TRUE)
dir <- if (length(args) >= 1) args[1] else file.path(dirname(sub("--file=", "", grep("--file=", commandArgs(), value = TRUE)[1])), "golden_bin")
shim <- Sys.getenv("VBNMF_SHIM", "")
if (nzchar(shim)) dyn.load(shim)
relerr <- function(a, b) max(abs(a - b) / pmax(abs(b), 1e-300))
read_case <- function(path) {
  dims <- read.table(file.path(path, "dims.txt"), stringsAsFactors = FALSE)
  out <- list()
  for (i in seq_len(nrow(dims))) {
    v <- readBin(file.path(path, paste0(dims[i, 1], ".f64")), what = "double", n = dims[i, 2] * dims[i, 3], size = 8, endian = "little")
    out[[dims[i, 1]]] <- matrix(v, nrow = dims[i, 2], ncol = dims[i, 3])
  }
  out
}
bad <- FALSE
for (case in sort(list.dirs(dir, full.names = FALSE, recursive = FALSE))) {
  g <- read_case(file.path(dir, case))
  r <- as.integer(g$r[1]); fudge <- g$fudge[1]
  hyper <- list(aw = g$hyper[1], bw = g$hyper[2], ah = g$hyper[3], bh = g$hyper[4])
  wh <- list(lw = g$lw0, lh = g$lh0, ew = g$lw0, eh = g$eh0)
  forms <- list(`R twin` = ccfindR:::vbnmf_updateR(g$X, wh, r, hyper, fudge),
                `native` = ccfindR:::vbnmf_update(g$X, wh, hyper, c(fudge)))
  if (nzchar(shim)) forms[["MI355X shim"]] <- .Call("_ccfindR_vbnmf_update", g$X, wh, hyper, c(fudge), PACKAGE = sub("\\.so$", "", basename(shim)))
  cat(sprintf("== %s (%d x %d, rank %d)\n", case, nrow(g$X), ncol(g$X), r))
  for (nm in names(forms)) {
    f <- forms[[nm]]
    errs <- sapply(c("lw", "lh", "ew", "eh", "dw", "dh"), function(k) relerr(f[[k]], g[[k]]))
    e_lkh <- abs(f$lkh / g$lkh[1] - 1)
    ok <- all(errs <= 1e-12) && e_lkh <= 1e-10
    bad <- bad || !ok
    cat(sprintf("   %-12s vs fixture: factors max rel %.2e (%s), lkh rel %.2e  %s\n", nm, max(errs), names(errs)[which.max(errs)], e_lkh,
                if (ok) "ok" else "EXCEEDS"))
  }
}
if (bad) quit(status = 1)
cat("all comparisons inside 1e-12 (factors) / 1e-10 (lkh)\n")
