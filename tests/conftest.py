import glob
import os
import sys


def _cpu_quota_threads():
    """Hardware threads the process may really use: those visible, capped by the container's cgroup CPU quota.  (The GPU boxes show 256
    threads under a 16-CPU quota; OpenMP pools of 256 spinning threads -- torch's, BLAS's, the oracle's -- get the whole process
    throttled, intermittently by a factor of 30: bench.py met it.)  Set before torch / numpy start their pools."""
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(-(-float(q) // float(per)))))
    except Exception:      # noqa: BLE001
        pass
    return max(1, n)


for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, str(_cpu_quota_threads()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(_cpu_quota_threads())

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built library: skip them (instead of failing at the first one) anywhere else."""
    lib = os.path.join(ROOT, "sigkernel_amd", "libsigkernel_amd.so")
    reason = None
    if not torch.cuda.is_available():
        reason = "no HIP device"
    elif not os.path.exists(lib):
        reason = "libsigkernel_amd.so has not been built"
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_gram_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "gram_*.npz")))


def reference_noise(name, key):
    """Round-off noise the REFERENCE's gradient `key` of fixture `name` carries: its max-norm relative distance from the
    reference's own formula evaluated in long double (tests/golden/grad_errors.json, written by
    tests/golden/measure_grad_noise.py, which imports no implementation)."""
    import json
    with open(os.path.join(GOLDEN, "grad_errors.json")) as f:
        return float(json.load(f)[name][key]["reference_noise"])


def grad_tol(name, key):
    """Tolerance for a gradient against the REFERENCE's fixture value: north_star's 1e-6, or -- for the few entries whose
    fixture itself is further than that from the reference's formula -- 1.25 x the reference's measured round-off noise.
    The reference differentiates by a forward difference with h = 1e-9 in double (sigkernel.py:313-341, :472-500);
    tests/test_oracle.py::test_every_gradient_fixture_vs_noise_free_reference_formula proves, per (fixture, key), that (i) the
    analytic adjoint is within 1e-7 of that formula in long double and (ii) the fixture's distance from it is the recorded
    noise -- so |implementation - fixture| <= noise + 1e-7 <= this tolerance, derived from the reference alone."""
    return max(1e-6, 1.25 * reference_noise(name, key))


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
