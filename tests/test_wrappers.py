"""Callers either side of the hot path: path transforms, MMD hypothesis test, SigCHSIC -- against
fixtures generated from the reference (tests/golden/make_golden.py::wrappers_case).  CPU run uses the oracle-backed
fake solver; the GPU run goes through the HIP kernels."""
import numpy as np
import pytest
import torch

import sigkernel_amd
from conftest import golden, rel_err


def test_transforms_match_the_reference():
    c = golden("wrappers")
    P = torch.from_numpy(c["paths"])
    for at in (0, 1):
        for ll in (0, 1):
            got = sigkernel_amd.transform(P, at=bool(at), ll=bool(ll), scale=0.5)
            want = c["transform_at%d_ll%d" % (at, ll)]
            assert got.shape == want.shape
            assert np.array_equal(got.numpy(), want)
    assert sigkernel_amd.lead_lag(P).shape == (4, 17, 4) and sigkernel_amd.add_time(P).shape == (4, 9, 3)


def _check_stats(dev):
    c = golden("wrappers")
    X, Y, Z = (torch.from_numpy(c[k]).to(dev) for k in ("X", "Y", "Z"))
    k = sigkernel_amd.RBFKernel(1.0)
    chsic = sigkernel_amd.SigCHSIC(X, Y, Z, k, dyadic_order=1, eps=0.1)
    assert abs(float(chsic) - float(c["chsic"])) <= 1e-9 * max(1.0, abs(float(c["chsic"])))
    rejected, stat, thr = sigkernel_amd.hypothesis_test(X, Y, k, confidence_level=0.99, dyadic_order=0, verbose=False)
    assert abs(float(stat) - float(c["mmd_d0"])) <= 1e-11
    assert abs(float(thr) - float(c["c_alpha"])) <= 1e-15 and sigkernel_amd.c_alpha(6, 0.99) == float(c["c_alpha"])
    assert int(rejected) == int(c["verdict_rejected"])


def test_stats_wrappers_on_cpu_with_the_fake_backend(oracle_backend):
    _check_stats("cpu")


@pytest.mark.gpu
def test_stats_wrappers_on_gpu():
    _check_stats("cuda:0")


@pytest.mark.gpu
def test_transforms_on_gpu_feed_the_kernel():
    c = golden("wrappers")
    P = torch.from_numpy(c["paths"]).to("cuda:0")
    Q = sigkernel_amd.transform(P, at=True, ll=True, scale=0.5)
    assert Q.device == P.device and np.array_equal(Q.cpu().numpy(), c["transform_at1_ll1"])
    K = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1).compute_Gram(Q, Q, sym=True)
    assert rel_err(K.cpu().numpy(), K.t().cpu().numpy()) <= 1e-12
