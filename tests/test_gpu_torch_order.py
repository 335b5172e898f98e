"""The library and PyTorch in one process, library used first: torch ships its own HIP runtime, and a process that maps
/opt/rocm's copy (through this library) before torch's ends with two of them -- torch.cuda then reports "No HIP GPUs are
available".  ccfindr_amd._native.load() therefore maps torch's first when torch is installed.  Run in a fresh
interpreter, where nothing has imported torch yet (pytest's collection of the other test modules does)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, numpy as np
assert "torch" not in sys.modules
import ccfindr_amd as C
X = np.random.default_rng(0).poisson(0.5, size=(60, 80)).astype(np.float64) + np.eye(60, 80)
M = C.CountMatrix(X)
eng = C.VBEngine(M, 3)                      # the library initialises HIP here
import torch
t = torch.ones(4, device="cuda")            # and torch afterwards, on the same runtime
assert float(t.sum().item()) == 4.0
red = C.VBEngine(M, 3, cols=(0, 40), m_global=80).reduce_tensor()    # engine memory as a torch tensor
assert red.is_cuda and red.dtype == torch.float64
print("ok")
"""


def test_library_first_then_torch_share_one_hip_runtime():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-1500:])
