"""A numpy engine with the HIP engine's surface and phase structure -- TEST INFRASTRUCTURE.

It restates, on the CPU, the algebra the device engine uses (not the reference's dense
statement order): sufficient statistics carried from the previous sweep, the collapsed
-sum(ew*eh), and the identity
    sum_ij X_ij (A+B)_ij / wth_ij = sum_ik sw_ik log lw_ik + sum_kj sh_kj log lh_kj
(A, B of reference src/vbnmf_update.cpp:69-72) that lets one pass over X give the statistics of
step t+1 and the evidence of step t.  tests/test_host_logic.py checks it against the literal
oracle, then uses it to drive the product's host loop and the cell-partition protocol without a GPU.
"""
import numpy as np
from scipy.special import digamma, gammaln

EPS = float(np.finfo(np.float64).eps)


class NumpyPhaseEngine:
    def __init__(self, X, rank, cols=None, m_global=None):
        X = np.asarray(X.toarray() if hasattr(X, "toarray") else X, dtype=np.float64)
        self.n, mfull = X.shape
        cb, ce = cols if cols is not None else (0, mfull)
        self.X = X[:, cb:ce]
        self.m = ce - cb
        self.m_global = mfull if m_global is None else m_global
        self.rank = int(rank)
        self.partitioned = self.m != self.m_global
        self.lgx = float(gammaln(self.X[self.X != 0] + 1.0).sum())
        self.red = np.zeros(self.n * self.rank + self.rank + 4)
        self._pending = False
        self.closed = False

    # ---- pieces ----------------------------------------------------------------------
    def _sweep(self):
        """Statistics of (lw, lh) on the stored entries, and this partition's data term."""
        X, lw, lh = self.X, self.lw, self.lh
        nz = X != 0
        wth = lw @ lh
        q = np.where(nz, X / np.where(nz, wth, 1.0), 0.0)
        self.swacc = q @ lh.T                          # sw = lw * swacc
        self.shacc = lw.T @ q                          # sh = lh * shacc
        xlog = float((X[nz] * np.log(wth[nz])).sum())
        data = float((self.swacc * lw * np.log(lw)).sum() + (self.shacc * lh * np.log(lh)).sum() - xlog)
        return data

    def _pack(self, data, UH, slh):
        nr = self.n * self.rank
        self.red[:nr] = self.swacc.ravel()
        self.red[nr:nr + self.rank] = self.eh.sum(axis=1)
        self.red[nr + self.rank:] = (UH, slh, data, self.lgx)

    # ---- VBEngine surface ----------------------------------------------------------------
    def set_state(self, lw, lh, eh):
        self.lw, self.lh, self.eh = (np.array(a, dtype=np.float64) for a in (lw, lh, eh))
        assert self.lw.shape == (self.n, self.rank) and self.lh.shape == (self.rank, self.m)
        self.ew = np.zeros_like(self.lw); self.dw = np.zeros_like(self.lw); self.dh = np.zeros_like(self.lh)
        self._pack(self._sweep(), 0.0, 0.0)
        self._state_pending = True
        if not self.partitioned:
            self.state_finish()

    def state_finish(self):
        assert self._state_pending
        self._state_pending = False

    def step_local(self, hyper, fudge=EPS):
        aw, bw, ah, bh = (float(hyper[k]) for k in ("aw", "bw", "ah", "bh"))
        n, r = self.n, self.rank
        nr = n * r
        sw = self.lw * self.red[:nr].reshape(n, r)              # global gene statistics
        rowsum_eh = self.red[nr:nr + r]                         # global, of the incoming eh
        alw = aw + sw
        bew = aw / bw + rowsum_eh
        self.ew = alw / bew
        self.dw = alw / bew / bew
        tmp = np.exp(digamma(alw)) / bew
        self.lw = np.where(tmp > fudge, tmp, fudge)
        self._UW = float((-(aw / bw) * self.ew + (-gammaln(aw) + aw * np.log(aw / bw)) + alw * (1 - np.log(bew)) + gammaln(alw)).sum())
        self._slw = float(np.log(self.lw).sum())
        self._csew = self.ew.sum(axis=0)
        sh = self.lh * self.shacc                               # local cell statistics
        alh = ah + sh
        beh = (ah / bh + self._csew)[:, None]
        self.eh = alh / beh
        self.dh = alh / beh / beh
        tmp = np.exp(digamma(alh)) / beh
        self.lh = np.where(tmp > fudge, tmp, fudge)
        UH = float((-(ah / bh) * self.eh + (-gammaln(ah) + ah * np.log(ah / bh)) + alh * (1 - np.log(beh)) + gammaln(alh)).sum())
        slh = float(np.log(self.lh).sum())
        self._pack(self._sweep(), UH, slh)
        self._pending = True

    def step_finish(self):
        assert self._pending
        self._pending = False
        n, r, mg = self.n, self.rank, self.m_global
        nr = n * r
        rs = self.red[nr:nr + r]
        UH, slh, data, lgx = self.red[nr + r:]
        U = -float(self._csew @ rs) - data - lgx + self._UW + UH
        lkh = U / (float(n) * float(mg))
        stats = (self._slw / (n * r), slh / (mg * r), float(self._csew.sum()) / (n * r), float(rs.sum()) / (mg * r))
        return lkh, stats

    def step(self, hyper, fudge=EPS):
        assert not self.partitioned
        self.step_local(hyper, fudge)
        return self.step_finish()

    def reduce_tensor(self):
        import torch
        return torch.from_numpy(self.red)

    def get_state(self, names=("lw", "lh", "ew", "eh", "dw", "dh")):
        return {k: np.array(getattr(self, k)) for k in names}

    def close(self):
        self.closed = True
