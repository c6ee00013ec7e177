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


def test_transformer_classes_match_the_reference_interface():
    """AddTime / LeadLag as classes (transformers.py:30-44, :57-80): fit / transform / fit_transform / transform_instance on lists
    of per-path arrays (ragged lengths included) and on one batched tensor; values pinned by the `transform` fixtures."""
    c = golden("wrappers")
    P = c["paths"]
    ll = sigkernel_amd.LeadLag().fit_transform(list(0.5 * P))
    assert isinstance(ll, list) and np.array_equal(np.array(ll), c["transform_at0_ll1"])
    at = sigkernel_amd.AddTime().fit_transform(ll)
    assert np.array_equal(np.array(at), c["transform_at1_ll1"])
    assert np.array_equal(np.array(sigkernel_amd.AddTime().fit(None).transform(list(0.5 * P))), c["transform_at1_ll0"])
    # tensors stay tensors, in one batched op
    T = sigkernel_amd.AddTime().transform(sigkernel_amd.LeadLag().transform(torch.from_numpy(0.5 * P)))
    assert isinstance(T, torch.Tensor) and np.array_equal(T.numpy(), c["transform_at1_ll1"])
    # ragged input, 1-d paths, constructor arguments
    rag = [np.arange(5.0), np.arange(3.0) * 2]
    out = sigkernel_amd.AddTime(init_time=2.0, total_time=9.0).transform(rag)
    assert out[0].shape == (5, 2) and out[1].shape == (3, 2) and np.array_equal(out[1][:, 0], np.linspace(2.0, 3.0, 3))
    assert sigkernel_amd.AddTime(init_time=2.0).get_params()["init_time"] == 2.0
    lead = sigkernel_amd.LeadLag().transform_instance(np.array([[1.0], [2.0], [4.0]]))
    assert np.array_equal(lead, np.array([[1, 1], [1, 2], [2, 2], [2, 4], [4, 4]], dtype=float))


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
