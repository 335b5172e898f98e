"""A CPU stand-in with the ML surface of VBEngine (ml_set_state / ml_step / ml_get_state / ml_likelihood / close),
driven by the oracle -- TEST INFRASTRUCTURE.  tests/test_host_factorize.py uses it to run the product's
factorize() loop (ccfindr_amd/factorize.py) without a GPU."""
import numpy as np

from oracle import mlnmf_oracle as O


class OracleMLEngine:
    def __init__(self, X, rank):
        self.X = np.asarray(X.toarray() if hasattr(X, "toarray") else X, dtype=np.float64)
        self.rank = int(rank)
        self.n, self.m = self.X.shape
        self.closed = False

    def ml_set_state(self, w, h):
        self.w, self.h = np.array(w, dtype=np.float64), np.array(h, dtype=np.float64)
        self.lk = O.likelihood_literal(self.X, self.w, self.h)

    def ml_step(self, prior=False, gamma_a=1.0, gamma_b=1.0):
        o = O.nmf_update_literal(self.X, self.w, self.h, prior, gamma_a, gamma_b)
        self.w, self.h = o["ew"], o["eh"]
        self.lk = O.likelihood_literal(self.X, self.w, self.h)
        return self.lk

    def ml_likelihood(self):
        return self.lk

    def ml_get_state(self, names=("ew", "eh")):
        return {k: v.copy() for k, v in (("ew", self.w), ("eh", self.h)) if k in names}

    def close(self):
        self.closed = True
