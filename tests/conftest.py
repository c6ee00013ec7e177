import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_gram_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "gram_*.npz")))


def make_kernel(case):
    import sigkernel_amd
    name = str(case["kernel"])
    return sigkernel_amd.LinearKernel() if name == "linear" else sigkernel_amd.RBFKernel(float(case["param"]))


def rel_err(a, b):
    """max-norm relative error, the parity measure of SURVEY 8(d)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture
def oracle_backend():
    """Install the oracle-backed fake solver back-end (host-logic tests on CPU); restore afterwards."""
    from sigkernel_amd import _lib
    from fake_backend import OracleBackend
    prev = _lib.set_backend(OracleBackend())
    yield
    _lib.set_backend(prev)


def walk(gen, A, M, D, dtype=torch.float64):
    return (torch.cumsum(torch.randn(A, M, D, generator=gen, dtype=torch.float64), dim=1) / np.sqrt(M * D)).to(dtype)
