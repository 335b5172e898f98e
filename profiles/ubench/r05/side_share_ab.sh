#!/bin/bash
# side_share_ab.sh: engine creation at the headline size with the two sides cut side by side, each side's parallel_for calls on
# all host threads (default) or on half of them (VBNMF_SIDE_THREAD_SHARE=2), and one after the other; interleaved, fresh processes.
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench
name, X, r = bench.make_workload(False)
X = X.tocsc()
np.savez("/tmp/vbnmf_c3.npz", data=X.data, indices=X.indices, indptr=X.indptr, shape=np.asarray(X.shape))
PY
for rep in 1 2 3; do
  python3 profiles/ubench/r05/setup_times.py "side by side, all threads each"
  VBNMF_SIDE_THREAD_SHARE=2 python3 profiles/ubench/r05/setup_times.py "side by side, half each"
  VBNMF_SERIAL_SIDES=1 python3 profiles/ubench/r05/setup_times.py "one after the other"
done
