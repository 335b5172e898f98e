"""10x-format input / output around the engine's native Matrix Market reader: mirror of the reference's
``read_10x`` (R/utils.R:28-54), ``write_10x`` (R/utils.R:867-884) and ``remove_zeros``, without the
SingleCellExperiment container (out of scope): counts stay sparse from the file to the device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from .engine import CountMatrix


@dataclass
class CountData:
    """What the reference keeps in an scNMFSet after read_10x: the counts and the two annotation tables."""
    counts: object                                   # scipy.sparse.csc_matrix, genes x cells
    genes: list = field(default_factory=list)        # rows of genes.tsv, each a list of columns; [0] is the row name
    barcodes: list = field(default_factory=list)     # rows of barcodes.tsv

    @property
    def rownames(self):
        return [g[0] for g in self.genes]

    @property
    def colnames(self):
        return [b[0] for b in self.barcodes]

    def count_matrix(self):
        """The ingested form the engines take."""
        return CountMatrix(self.counts)


def _read_table(path):
    """``utils::read.table(path, stringsAsFactors = FALSE)``: whitespace-separated columns, no header."""
    rows = []
    with open(path) as f:
        for line in f:
            a = line.split()
            if a:
                rows.append(a)
    return rows


def remove_zeros(x: CountData) -> CountData:
    """Drop all-zero rows and columns (what ``remove_zeros`` does to the object, R/utils.R:52)."""
    X = x.counts
    keep_r = np.flatnonzero(np.asarray(X.sum(axis=1)).ravel() > 0)
    keep_c = np.flatnonzero(np.asarray(X.sum(axis=0)).ravel() > 0)
    if len(keep_r) == X.shape[0] and len(keep_c) == X.shape[1]:
        return x
    X = X[keep_r][:, keep_c].tocsc()
    return CountData(X, [x.genes[i] for i in keep_r] if x.genes else [], [x.barcodes[j] for j in keep_c] if x.barcodes else [])


def read_10x(dir, count="matrix.mtx", genes="genes.tsv", barcodes="barcodes.tsv", remove_zeros_=True):
    """``read_10x(dir, count, genes, barcodes, remove.zeros)``; reference R/utils.R:28-54."""
    if not os.path.isdir(dir):
        raise FileNotFoundError(f"Input directory {dir} does not exist")            # :31
    cpath = os.path.join(dir, count)
    if not os.path.exists(cpath):
        raise FileNotFoundError(f"Count file {cpath} does not exist")               # :33
    M = CountMatrix.from_mtx(cpath)                                                 # :34
    try:
        X = M.to_scipy()
    finally:
        M.close()
    gpath = os.path.join(dir, genes)
    if not os.path.exists(gpath):
        raise FileNotFoundError(f"Count file {gpath} does not exist")               # :36-37 (the reference's wording)
    glist = _read_table(gpath)                                                      # :38
    bpath = os.path.join(dir, barcodes)
    if not os.path.exists(bpath):
        raise FileNotFoundError(f"Count file {bpath} does not exist")               # :40-41
    clist = _read_table(bpath)                                                      # :42
    if len(glist) != X.shape[0] or len(clist) != X.shape[1]:
        raise ValueError("annotation tables do not match the count matrix")        # dimnames<- fails in R (:43-44)
    x = CountData(X, glist, clist)
    return remove_zeros(x) if remove_zeros_ else x                                  # :52


def write_10x(x: CountData, dir, count="matrix.mtx", genes="genes.tsv", barcodes="barcodes.tsv", quote=False):
    """``write_10x(object, dir, count, genes, barcodes, quote)``; reference R/utils.R:867-884."""
    M = CountMatrix(x.counts)
    try:
        M.write_mtx(os.path.join(dir, count))                                       # :876
    finally:
        M.close()
    q = (lambda s: f'"{s}"') if quote else (lambda s: s)
    with open(os.path.join(dir, genes), "w") as f:                                  # :879-880
        for g in x.genes:
            f.write(" ".join(q(c) for c in g) + "\n")
    with open(os.path.join(dir, barcodes), "w") as f:                               # :881-882 (never quoted)
        for b in x.barcodes:
            f.write(" ".join(b) + "\n")
    return x
