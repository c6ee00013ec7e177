"""BASELINE.json's big configurations through the public API on the GPU, at sizes the CPU oracle can follow:

* C5 (batch 256, len 512, dim 16, RBFKernel, dyadic 2, fp32): the exact workload at reduced batch;
* C4 (2048 x 2048, len 64, dim 4, RBFKernel, dyadic 2, compute_mmd + backward): one row shard (8 rows of X) against all
  2048 paths of Y at full length -- what one rank of the sharded job computes;
* the documented reference-side binding, executed verbatim;
* the example pipeline of SURVEY 8(f) #4 (transform -> compute_Gram(sym=True) -> SVC(kernel='precomputed'));
* failure-injection for the fused-adjoint fallback and NaN propagation on both forward routes.
"""
import os
import re

import numpy as np
import pytest
import torch

import sigkernel_amd
from sigkernel_amd import _lib
from sigkernel_amd import sigkernel as skmod
from conftest import ROOT, golden, grad_tol, golden_gram_cases, make_kernel, rel_err, walk
from oracle import oracle as O

DEV = "cuda:0"
def _usable_threads():
    """Hardware threads capped by the container's cgroup CPU quota (the GPU boxes show 256 threads under a 16-CPU quota: an
    oversubscribed OpenMP team spinning at its barriers was measured to slow the oracle down 30x, intermittently)."""
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(-(-float(q) // float(per)))))
    except Exception:      # noqa: BLE001
        pass
    return max(1, n)


NT = min(32, _usable_threads())
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

# fp32 I/O with the PDE state in fp64: the reference's own fp32 bar (sigkernel/test_mps.py:32)
F32_RTOL, F32_ATOL = 1e-4, 1e-5


@pytest.mark.gpu
def test_c5_workload_at_reduced_batch():
    """C5: len 512, dim 16, RBFKernel(1.0), dyadic 2, fp32 tensors -- 4 x 4 pairs of the 2044 x 2044 grid, against the fp64
    oracle on the up-cast inputs.  Stated tolerance: rtol 1e-4 / atol 1e-5 (the reference's fp32 acceptance); the achieved
    error is far smaller because only I/O is fp32, and is asserted at 2e-6 so that a regression to an fp32 PDE state
    (5e-3 at this grid size) cannot hide."""
    gen = torch.Generator().manual_seed(55)
    Xc, Yc = walk(gen, 4, 512, 16, torch.float32), walk(gen, 4, 512, 16, torch.float32)
    k = sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, dyadic_order=2)
    X, Y = Xc.to(DEV), Yc.to(DEV)
    K = sk.compute_Gram(X, Y)
    assert K.dtype == torch.float32 and K.shape == (4, 4)
    want = O.gram_forward(Xc.double(), Yc.double(), k, 2, nthreads=NT)
    np.testing.assert_allclose(K.cpu().numpy(), want, rtol=F32_RTOL, atol=F32_ATOL)
    assert rel_err(K.cpu().numpy(), want) <= 2e-6
    # paired and symmetric entry points at the same shape
    Kp = sk.compute_kernel(X, Y)
    np.testing.assert_allclose(Kp.cpu().numpy(), np.diag(want), rtol=F32_RTOL, atol=F32_ATOL)
    Ks = sk.compute_Gram(X, X, sym=True)
    assert torch.equal(Ks, Ks.t())
    np.testing.assert_allclose(Ks.cpu().numpy(), O.gram_forward(Xc.double(), Xc.double(), k, 2, nthreads=NT), rtol=F32_RTOL,
                               atol=F32_ATOL)
    # the adjoint PDE on the same grids (2 x 2 pairs): weighted-sum gradient against the oracle's closed form
    w = torch.tensor([[1.0, -0.5], [0.25, 2.0]])
    Xg = X[:2].clone().requires_grad_(True)
    (sk.compute_Gram(Xg, Y[:2]) * w.to(DEV)).sum().backward()
    gwant = O.gram_grad_weighted(Xc[:2].double(), Yc[:2].double(), w.numpy(), k, 2, nthreads=NT)
    assert Xg.grad.dtype == torch.float32
    assert rel_err(Xg.grad.cpu().numpy(), gwant) <= 2e-4


@pytest.mark.gpu
def test_c5_workload_fp64_tensors_same_shape():
    """The same shape with fp64 tensors meets the fp64 bar (north_star: 1e-6; achieved ~1e-12)."""
    gen = torch.Generator().manual_seed(56)
    Xc, Yc = walk(gen, 3, 512, 16), walk(gen, 2, 512, 16)
    k = sigkernel_amd.RBFKernel(1.0)
    K = sigkernel_amd.SigKernel(k, 2).compute_Gram(Xc.to(DEV), Yc.to(DEV))
    assert rel_err(K.cpu().numpy(), O.gram_forward(Xc, Yc, k, 2, nthreads=NT)) <= 1e-10


@pytest.mark.gpu
def test_c4_row_shard_at_full_length():
    """C4: one rank's share of the sharded job -- 8 rows of X against ALL 2048 paths of Y (len 64, dim 4, RBFKernel(1.0),
    dyadic 2, fp64), compute_mmd(X, Y).backward().  The 8 x 2048 block and the gradient are checked in full against the
    oracle's closed form; the 2048 x 2048 block K_YY (4.2 M pairs, no gradient) is spot-checked on 64 entries and must be
    exactly symmetric; the MMD value is re-assembled from the oracle's blocks and the GPU's K_YY sum."""
    gen = torch.Generator().manual_seed(44)
    A, B, M, D, d = 8, 2048, 64, 4, 2
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, M, D)
    k = sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, dyadic_order=d)
    X, Y = Xc.to(DEV), Yc.to(DEV)
    Xg = X.clone().requires_grad_(True)
    mmd = sk.compute_mmd(Xg, Y)
    mmd.backward()
    # forward blocks
    Kxy = sk.compute_Gram(X, Y)
    want_xy = O.gram_forward(Xc, Yc, k, d, nthreads=NT)
    assert rel_err(Kxy.cpu().numpy(), want_xy) <= 1e-11
    want_xx = O.gram_forward(Xc, Xc, k, d, nthreads=NT)
    Kyy = sk.compute_Gram(Y, Y, sym=True)
    assert torch.equal(Kyy, Kyy.t())
    rng = np.random.default_rng(3)
    for p in rng.integers(0, B * B, size=64):
        i, j = divmod(int(p), B)
        assert abs(float(Kyy[i, j]) - O.gram_forward(Yc[i:i + 1], Yc[j:j + 1], k, d)[0, 0]) <= 1e-11 * abs(float(Kyy[i, j]))
    kyy_m = (float(Kyy.sum()) - float(torch.diag(Kyy).sum())) / (B * (B - 1.0))
    want_mmd = (want_xx.sum() - np.trace(want_xx)) / (A * (A - 1.0)) + kyy_m - 2.0 * want_xy.mean()
    assert abs(float(mmd.detach()) - want_mmd) <= 1e-10
    # gradient: d/dX [ sum_offdiag K_XX / (A (A-1)) ] under the reference's 2x rule (sigkernel.py:410-412) - 2 mean K_XY
    w_xx = (np.ones((A, A)) - np.eye(A)) / (A * (A - 1.0))
    g_xx = 2.0 * O.gram_grad_weighted(Xc, Xc, w_xx, k, d, nthreads=NT)
    g_xy = O.gram_grad_weighted(Xc, Yc, np.full((A, B), -2.0 / (A * B)), k, d, nthreads=NT)
    assert rel_err(Xg.grad.cpu().numpy(), g_xx + g_xy) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_gram_cases())
def test_api_gradients_against_the_oracle_closed_form(name):
    """The per-fixture tolerances against the reference are set by the reference's finite-difference noise (1e-6 .. 2e-5);
    against the oracle's analytic closed form the HIP path is held to 1e-9 on every fixture, so a regression of that size
    cannot hide under the looser bound."""
    c = golden(name)
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]))
    Xg = X.to(DEV).requires_grad_(True)
    (sk.compute_Gram(Xg, Y.to(DEV)) * w.to(DEV)).sum().backward()
    want = O.gram_grad_weighted(X, Y, w.numpy(), make_kernel(c), int(c["dyadic"]), bool(c["naive"]), nthreads=8)
    assert rel_err(Xg.grad.cpu().numpy(), want) <= 1e-9
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_w"]) <= grad_tol(name, "grad_w")


def test_oracle_weighted_gradient_is_the_contraction_of_grad_points():
    c = golden("gram_c2mini_rbf_d1")
    X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
    gp = O.gram_grad_points(X, Y, make_kernel(c), 1)
    gw = O.gram_grad_weighted(X, Y, w.numpy(), make_kernel(c), 1)
    assert rel_err(gw, np.einsum("ab,abmd->amd", w.numpy(), gp)) <= 1e-13


# ---------------------------------------------------------------------------------------------
# the binding a reference maintainer would add, executed as documented
# ---------------------------------------------------------------------------------------------
def _integration_snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = [b for b in blocks if "def sigkernel_Gram_hip" in b]
    assert len(code) == 1, "INTEGRATION.md section B must hold exactly one sigkernel_Gram_hip snippet"
    return code[0]


def test_integration_snippet_binds_declared_symbols_only():
    """CPU leg: every sk_* name the snippet touches is declared in include/sigkernel_amd.h."""
    header = open(os.path.join(ROOT, "include", "sigkernel_amd.h")).read()
    names = set(re.findall(r"_lib\.(sk_\w+)", _integration_snippet()))
    assert names and all(re.search(r"\b%s\(" % n, header) for n in names), names


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gram_c2mini_rbf_d1", "gram_c3mini_lin_d1", "gram_rbf_d1_naive", "gram_lin_d2_ragged"])
def test_integration_snippet_runs_verbatim(name):
    """The documented `sigkernel_Gram_hip` (ctypes over the C ABI, fed with the static Gram matrix exactly like the
    reference's 'cuda' branch, sigkernel.py:362-382) reproduces the reference's Gram matrix on the golden fixtures."""
    code = _integration_snippet().replace('ctypes.CDLL("libsigkernel_amd.so")', 'ctypes.CDLL(%r)' % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    c = golden(name)
    X, Y = torch.from_numpy(c["X"]).to(DEV), torch.from_numpy(c["Y"]).to(DEV)
    G_static = make_kernel(c).Gram_matrix(X, Y).contiguous()
    K = ns["sigkernel_Gram_hip"](G_static, int(c["dyadic"]), bool(c["naive"]))
    torch.cuda.synchronize()
    assert rel_err(K.cpu().numpy(), c["gram"]) <= 1e-11


# ---------------------------------------------------------------------------------------------
# the example pipeline
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_example_classification_pipeline():
    """transform(at) -> compute_Gram(sym=True) -> GridSearchCV(SVC(kernel='precomputed')) -> test Gram -> predict, as
    examples/time_series_classification.py:94, :189-202, :262-281 of the reference, on synthetic two-class paths."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sk_example", os.path.join(ROOT, "examples", "time_series_classification.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    x_train, y_train = ex.make_dataset(60, 40, seed=0)
    x_test, y_test = ex.make_dataset(40, 40, seed=1)
    score, sigma, model, xt = ex.fit_signature_svc(x_train, y_train, torch.device(DEV), sigmas=(0.5, 1.0), cv=3)
    pred = ex.predict(model, sigma, xt, x_test, np.abs(x_train).max(), torch.device(DEV))
    assert score >= 0.9 and float(np.mean(pred == y_test)) >= 0.9
    # the Gram matrix the classifier was trained on is the oracle's
    k = sigkernel_amd.RBFKernel(sigma)
    G = sigkernel_amd.SigKernel(k, 0).compute_Gram(xt[:6], xt[:6], sym=True)
    assert rel_err(G.cpu().numpy(), O.gram_forward(xt[:6].cpu(), xt[:6].cpu(), k, 0)) <= 1e-11


# ---------------------------------------------------------------------------------------------
# failure injection
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fused_adjoint_failure_falls_back_tiled_by_the_unfused_budget(monkeypatch):
    """When the fused linear adjoint declines a case the backward pass must take the unfused route TILED BY THAT ROUTE'S
    transient memory (3 arrays of the tile's increments), not in the one tile the fused kernel was sized for."""
    gen = torch.Generator().manual_seed(9)
    X, Y = walk(gen, 24, 40, 5).to(DEV), walk(gen, 16, 33, 5).to(DEV)
    w = torch.randn(24, 16, generator=gen, dtype=torch.float64).to(DEV)
    budget = 3 * 16 * 40 * 33 * 8 * 5          # room for 5 rows of the unfused route
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1, workspace_bytes=budget)
    Xr = X.clone().requires_grad_(True)
    (sk.compute_Gram(Xr, Y) * w).sum().backward()            # reference run: fused adjoint, one launch
    be = _lib.get_backend()
    orig = type(be).linear_adjoint_fused

    def failing(self, *a, **k):
        orig(self, *a, **k)
        return None

    tiles = []
    orig_tile = skmod._tile_gradient
    monkeypatch.setattr(type(be), "linear_adjoint_fused", failing)
    monkeypatch.setattr(skmod, "_tile_gradient", lambda be_, sk_, Xt, *a, **k: (tiles.append(Xt.shape[0]), orig_tile(be_, sk_, Xt, *a, **k))[1])
    Xg = X.clone().requires_grad_(True)
    (sk.compute_Gram(Xg, Y) * w).sum().backward()
    assert len(tiles) >= 5 and max(tiles) <= 5 and sum(tiles) == 24, tiles
    assert rel_err(Xg.grad.cpu().numpy(), Xr.grad.cpu().numpy()) <= 1e-10
    # paired batches take the same fallback
    tiles.clear()
    Xp = X[:16].clone().requires_grad_(True)
    sk.compute_kernel(Xp, Y).sum().backward()
    assert tiles and sum(tiles) == 16 and torch.isfinite(Xp.grad).all()


@pytest.mark.gpu
def test_adjoint_rescue_runs_on_the_device_without_a_host_sync():
    """solve_adj never reads the residuals back: the stored-grid re-solve of flagged pairs is a kernel of its own.  Run under
    torch's sync-debug mode, which raises on any operation that synchronises the host with the device."""
    be = _lib.HipBackend()
    rng = np.random.default_rng(5)
    inc = rng.normal(scale=0.02, size=(70, 63, 63))
    inc[[2, 65]] = rng.normal(scale=0.9, size=(2, 63, 63))      # wild pairs in two different 64-pair scan windows
    ld = _lib._padded_ld(63, 8)
    buf = torch.zeros(70, 63, ld, dtype=torch.float64, device=DEV)
    buf[..., :63] = torch.from_numpy(inc).to(DEV)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        k, W, res = be.solve_adj(buf[..., :63], 1, return_residual=True)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    want_k, want_w = O.adjoint_coarse(inc, 1, nthreads=8)
    res = res.cpu().numpy()
    assert res[2] > be.ADJ_RESIDUAL_TOL and res[65] > be.ADJ_RESIDUAL_TOL
    for p in (2, 65):
        assert rel_err(W.cpu().numpy()[p], want_w[p]) <= 1e-12          # re-solved by the bit-exact kernel
        assert abs(float(k[p]) - want_k[p]) <= 1e-12 * abs(want_k[p])
    keep = np.delete(np.arange(70), [2, 65])
    assert rel_err(W.cpu().numpy()[keep], want_w[keep]) <= 1e-10


@pytest.mark.gpu
def test_backward_passes_never_synchronise():
    """no hidden synchronisation.  A training step -- forward with edges, adjoint, static-kernel chain rule -- must not
    synchronise at all, on any of the three backward routes: the unfused one (RBF, dim 6: flagged pairs are re-solved by
    sk_adj_rescue_*), the fused RBF adjoint (dim 4) and the fused linear adjoint, whose exploding pairs are rescued on the device
    too (sk_adj_fused_rescue.hip) instead of being looked at by the host."""
    import warnings
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(12, 9, generator=gen, dtype=torch.float64).to(DEV)
    for kern, D in ((sigkernel_amd.RBFKernel(1.0), 6), (sigkernel_amd.RBFKernel(1.0), 4), (sigkernel_amd.LinearKernel(), 4)):
        X, Y = walk(gen, 12, 40, D).to(DEV), walk(gen, 9, 33, D).to(DEV)
        sk = sigkernel_amd.SigKernel(kern, 1)
        Xg = X.clone().requires_grad_(True)
        (sk.compute_Gram(Xg, Y) * w).sum().backward()        # warm-up: library load, allocator
        sk.compute_mmd(X[:9].clone().requires_grad_(True), Y).backward()
        torch.cuda.synchronize()
        Xg = X.clone().requires_grad_(True)
        Xm = X[:9].clone().requires_grad_(True)
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                (sk.compute_Gram(Xg, Y) * w).sum().backward()
                sk.compute_kernel(X[:9], Y)
                sk.compute_mmd(Xm, Y).backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        syncs = [r for r in rec if "synchroniz" in str(r.message).lower()]
        assert len(syncs) == 0, (type(kern).__name__, [str(r.message) for r in syncs])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D", [("linear", 8), ("rbf", 4)])
def test_forward_and_training_step_replay_from_a_hip_graph(kind, D):
    """compute_Gram, and a whole compute_mmd(X, Y).backward(), captured into a hipGraph and replayed: proves there is no
    synchronisation, no host read-back and no allocation outside torch's caching allocator anywhere on the path (any of them
    aborts a capture), and that a replay reproduces the eager results bit for bit on fresh input values."""
    gen = torch.Generator().manual_seed(51)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, 1)
    X0, Y0 = walk(gen, 24, 40, D).to(DEV), walk(gen, 20, 33, D).to(DEV)
    X1, Y1 = walk(gen, 24, 40, D).to(DEV), walk(gen, 20, 33, D).to(DEV)
    sX, sY = X0.clone().requires_grad_(True), Y0.clone()

    def step():
        K = sk.compute_Gram(sX.detach(), sY)
        loss = sk.compute_mmd(sX, sY)
        loss.backward()
        return K, loss.detach()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):            # warm-up on a side stream, as torch's capture protocol asks
        for _ in range(3):
            step()
            sX.grad = None
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gK, gloss = step()
    ggrad = sX.grad
    for Xv, Yv in ((X1, Y1), (X0, Y0)):
        with torch.no_grad():
            sX.copy_(Xv)
            sY.copy_(Yv)
        graph.replay()
        torch.cuda.synchronize()
        Xe = Xv.clone().requires_grad_(True)
        Ke = sk.compute_Gram(Xe.detach(), Yv)
        le = sk.compute_mmd(Xe, Yv)
        le.backward()
        assert torch.equal(gK, Ke) and torch.equal(gloss, le.detach()) and torch.equal(ggrad, Xe.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,dt,A,B,M", [("rbf", 3, 1, torch.float64, 32, 32, 64), ("linear", 8, 1, torch.float64, 24, 40, 50), ("rbf", 4, 2, torch.float64, 40, 17, 30),
                                              ("rbf", 7, 0, torch.float64, 20, 20, 150), ("linear", 12, 1, torch.float64, 16, 16, 40), ("rbf", 3, 1, torch.float32, 32, 32, 64),
                                              ("rbf", 16, 2, torch.float32, 6, 5, 300)])
def test_merged_loss_route_on_the_gpu(kind, D, d, dt, A, B, M, monkeypatch):
    """Training-sized loss wrappers (sigkernel._SigKernelLoss: ONE forward launch with edges and ONE adjoint launch over K(X, [X; Y]))
    against the reference's composition (routes.no_merged_loss; sigkernel.py:146-197) and against the oracle's closed forms
    (O.gram_forward: _SigKernelGram.forward, :350-401; O.gram_grad_weighted: prep_backward + backward with the 2x rule, :404-502);
    captured into a hipGraph, a replay on fresh values reproduces the eager call bit for bit."""
    gen = torch.Generator().manual_seed(77)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.8)
    sk = sigkernel_amd.SigKernel(k, d)
    X, Y = walk(gen, A, M, D).to(dt).to(DEV), walk(gen, B, M, D).to(dt).to(DEV)
    f32 = dt == torch.float32
    for fn, Yv, yy in ((sk.compute_mmd, Y, True), (sk.compute_expected_scoring_rule, Y, False), (sk.compute_scoring_rule, Y[:1], False)):
        out = {}
        for composed in (False, True):
            monkeypatch.setattr(sigkernel_amd.routes, "no_merged_loss", composed)
            Xg = X.clone().requires_grad_(True)
            v = fn(Xg, Yv)
            v.backward()
            with torch.no_grad():
                v0 = fn(X, Yv)
            out[composed] = (float(v.detach()), Xg.grad.double().cpu().numpy(), float(v0))
        scale = max(1.0, abs(out[True][0]))
        assert abs(out[False][0] - out[True][0]) <= (2e-5 if f32 else 1e-12) * scale
        assert abs(out[False][2] - out[False][0]) <= (2e-5 if f32 else 1e-12) * scale
        assert rel_err(out[False][1], out[True][1]) <= (2e-4 if f32 else 1e-10)
        # the oracle: K(X, [X; Y]) and its gradient under the weights of the loss
        Xc, Yc = X.double().cpu(), Yv.double().cpu()
        Zc = torch.cat([Xc, Yc])
        Kxz = O.gram_forward(Xc, Zc, k, d, nthreads=NT)
        Bv = Yc.shape[0]
        wf = np.concatenate([(1.0 - np.eye(A)) / (A * (A - 1.0)), np.full((A, Bv), -2.0 / (A * Bv))], axis=1)
        want = float((Kxz * wf).sum())
        if yy:
            Kyy = O.gram_forward(Yc, Yc, k, d, nthreads=NT)
            want += float((Kyy.sum() - np.trace(Kyy)) / (Bv * (Bv - 1.0)))
        wb = wf.copy()
        wb[:, :A] *= 2.0
        gw = O.gram_grad_weighted(Xc, Zc, wb, k, d, nthreads=NT)
        assert abs(out[False][0] - want) <= (2e-5 if f32 else 1e-11) * max(1.0, abs(want))
        assert rel_err(out[False][1], gw) <= (2e-4 if f32 else 1e-9)
    # a captured training step
    monkeypatch.setattr(sigkernel_amd.routes, "no_merged_loss", False)
    sX, sY = X.clone().requires_grad_(True), Y.clone()

    def step():
        loss = sk.compute_mmd(sX, sY)
        loss.backward()
        return loss.detach()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
            sX.grad = None
    torch.cuda.current_stream().wait_stream(side)
    skmod._LOSS_WEIGHTS.clear()          # the constant weights built INSIDE the capture: nodes of the graph, not cached for eager calls
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gloss = step()
    assert not skmod._LOSS_WEIGHTS
    X1, Y1 = walk(gen, A, M, D).to(dt).to(DEV), walk(gen, B, M, D).to(dt).to(DEV)
    for Xv, Yv in ((X1, Y1), (X, Y)):
        with torch.no_grad():
            sX.copy_(Xv)
            sY.copy_(Yv)
        graph.replay()
        torch.cuda.synchronize()
        Xe = Xv.clone().requires_grad_(True)
        le = sk.compute_mmd(Xe, Yv)
        le.backward()
        assert torch.equal(gloss, le.detach()) and torch.equal(sX.grad, Xe.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,A,B,M,naive", [("rbf", 3, 1, 32, 32, 64, False), ("rbf", 3, 1, 64, 64, 64, False), ("linear", 8, 1, 24, 40, 50, False),
                                                  ("rbf", 4, 2, 40, 17, 30, False), ("rbf", 2, 0, 9, 2, 20, False), ("linear", 5, 0, 2, 3, 120, True),
                                                  ("rbf", 8, 1, 7, 5, 33, True), ("linear", 3, 2, 130, 3, 12, False), ("rbf", 1, 2, 3, 130, 9, False),
                                                  # big enough for the work queue: lanes enter drawn chunks, and the pairs that keep no edges
                                                  # (the triangle of K(Y, Y)) start in the middle of one
                                                  ("rbf", 3, 1, 320, 320, 33, False), ("linear", 6, 1, 300, 420, 40, False)])
def test_loss_launch_route_on_the_gpu(kind, D, d, A, B, M, naive, monkeypatch):
    """The one-launch glue of the loss wrappers (csrc/sk_loss.hip: sk_prep_cat, sk_solve_fwd_loss_f64 -- the rectangle K(X, [X; Y]) and
    the strict triangle of K(Y, Y) as ONE launch --, sk_loss_value, sk_loss_weights, sk_*_adjoint_finish) against the same merged
    route through torch ops (routes.no_loss_launch) and against the oracle's closed forms (O.gram_forward: sigkernel.py:350-401;
    O.gram_grad_weighted: :404-502 with the 2x rule); a second backward through the same graph gives the same bits."""
    gen = torch.Generator().manual_seed(91)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.7)
    sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
    X, Y = walk(gen, A, M, D).to(DEV), walk(gen, B, M, D).to(DEV)
    be = _lib.get_backend()
    for fn, Yv, yy in ((sk.compute_mmd, Y, True), (sk.compute_expected_scoring_rule, Y, False), (sk.compute_scoring_rule, Y[:1], False)):
        if yy and Yv.shape[0] < 2:
            continue
        out = {}
        for off in (False, True):
            monkeypatch.setattr(sigkernel_amd.routes, "no_loss_launch", off)
            Xg = X.clone().requires_grad_(True)
            v = fn(Xg, Yv)
            if not off:      # the route under test is the one taken
                assert skmod._loss_launch_ok(be, k, X, Yv, d, naive, True) is not None
            (g1,) = torch.autograd.grad(v, Xg, retain_graph=True)
            (g2,) = torch.autograd.grad(v, Xg)
            assert torch.equal(g1, g2)
            with torch.no_grad():
                v0 = fn(X, Yv)
            out[off] = (float(v.detach()), g1.cpu().numpy(), float(v0))
        scale = max(1.0, abs(out[True][0]))
        assert abs(out[False][0] - out[True][0]) <= 1e-12 * scale
        assert abs(out[False][2] - out[False][0]) <= 1e-12 * scale
        assert rel_err(out[False][1], out[True][1]) <= 1e-10
        Xc, Yc = X.cpu(), Yv.cpu()
        Zc = torch.cat([Xc, Yc])
        Kxz = O.gram_forward(Xc, Zc, k, d, naive=naive, nthreads=NT)
        Bv = Yc.shape[0]
        wf = np.concatenate([(1.0 - np.eye(A)) / (A * (A - 1.0)), np.full((A, Bv), -2.0 / (A * Bv))], axis=1)
        want = float((Kxz * wf).sum())
        if yy:
            Kyy = O.gram_forward(Yc, Yc, k, d, naive=naive, nthreads=NT)
            want += float((Kyy.sum() - np.trace(Kyy)) / (Bv * (Bv - 1.0)))
        wb = wf.copy()
        wb[:, :A] *= 2.0
        gw = O.gram_grad_weighted(Xc, Zc, wb, k, d, naive=naive, nthreads=NT)
        assert abs(out[False][0] - want) <= 1e-11 * max(1.0, abs(want))
        assert rel_err(out[False][1], gw) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rbf", "linear"])
def test_loss_launch_route_falls_back_when_the_fused_adjoint_declines(kind, monkeypatch):
    """the one-launch loss route's backward must not raise when the fused adjoint declines a shape sk_route_query routed to it
    (a scope or workspace check the query does not mirror): it falls back to the rows' gradient with the same weights -- the same
    gradient as the undisturbed route, to rounding."""
    gen = torch.Generator().manual_seed(17)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.8)
    sk = sigkernel_amd.SigKernel(k, 1)
    X, Y = walk(gen, 12, 30, 3).to(DEV), walk(gen, 9, 30, 3).to(DEV)
    be = _lib.get_backend()
    Xg = X.clone().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    want = Xg.grad.clone()
    name = "linear_adjoint_fused" if kind == "linear" else "rbf_adjoint_fused"
    real = getattr(type(be), name)
    calls = []

    def declining(self, *a, **kw):
        if kw.get("staged") is not None:      # the loss route's call (it hands over the arrays its forward staged)
            calls.append(1)
            return None
        return real(self, *a, **kw)
    # (on the TYPE: undoing a patch of the instance would leave the bound method in its __dict__, shadowing later patches of the type)
    monkeypatch.setattr(type(be), name, declining)
    Xg = X.clone().requires_grad_(True)
    sk.compute_mmd(Xg, Y).backward()
    assert calls, "the one-launch route was not the one taken"
    assert rel_err(Xg.grad.cpu().numpy(), want.cpu().numpy()) <= 1e-10


@pytest.mark.gpu
def test_loss_launch_route_checks_its_limits_and_budget_before_launching(monkeypatch):
    """pair fields of 15 / 31 bits are checked BEFORE any launch (the triangle of K(Y, Y) only where it is solved), and a call
    whose edges + values + weights + pair table would exceed keep_edges_fraction of the transient budget is left to the tiled route."""
    be = _lib.get_backend()
    k = sigkernel_amd.RBFKernel(1.0)
    gen = torch.Generator().manual_seed(3)
    X, Y = walk(gen, 8, 17, 2).to(DEV), walk(gen, 6, 17, 2).to(DEV)
    assert skmod._loss_launch_ok(be, k, X, Y, 1, False, True, True, None) is not None
    # a budget the edges of 8 x 14 pairs (8 (2 x 32 + 32) bytes each) cannot fit into: declined with a gradient pending, taken without
    assert skmod._loss_launch_ok(be, k, X, Y, 1, False, True, True, 4096) is None
    assert skmod._loss_launch_ok(be, k, X, Y, 1, False, False, True, 1 << 20) is not None
    # B = 47000 paths (sk_loss_value_f64 used to count B^2 >= 2^31 even without the triangle): too many for the 15-bit triangle field -> declined for the MMD, fine for the scoring rule
    Ybig = torch.zeros(47000, 3, 2, dtype=torch.float64, device=DEV)
    Xs = torch.zeros(4, 3, 2, dtype=torch.float64, device=DEV)
    assert skmod._loss_launch_ok(be, k, Xs, Ybig, 0, False, False, True, None) is None
    assert skmod._loss_launch_ok(be, k, Xs, Ybig, 0, False, False, False, None) is not None
    sk = sigkernel_amd.SigKernel(k, 0)
    v = sk.compute_expected_scoring_rule(Xs, Ybig)      # (ADVICE: B >= 46341 used to raise after the forward launch had run)
    assert torch.isfinite(v)
    # the constant weights of the torch-glue route: a bounded cache (least recently used first), pinned entries stay
    skmod._LOSS_WEIGHTS.clear()
    for a in range(2, 24):
        skmod._loss_weights(a, 3, torch.float64, X.device, max_cached=8)
    assert len(skmod._LOSS_WEIGHTS) == 8 and (23, 3, torch.float64, X.device) in skmod._LOSS_WEIGHTS and (2, 3, torch.float64, X.device) not in skmod._LOSS_WEIGHTS
    skmod._LOSS_WEIGHTS.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,A,B,M,N,naive", [("rbf", 3, 1, 9, 7, 300, 64, False), ("rbf", 4, 2, 5, 11, 129, 33, False), ("rbf", 2, 0, 6, 6, 500, 128, False),
                                                    ("rbf", 1, 1, 17, 3, 140, 20, True), ("rbf", 4, 0, 4, 9, 129, 128, False), ("rbf", 3, 2, 12, 12, 700, 64, True),
                                                    ("linear", 3, 1, 9, 7, 300, 64, False), ("linear", 8, 2, 5, 11, 129, 65, False),
                                                    ("linear", 2, 0, 6, 6, 500, 60, False), ("linear", 5, 1, 17, 3, 140, 20, True),
                                                    ("linear", 8, 0, 4, 9, 300, 70, False), ("linear", 1, 2, 12, 12, 700, 40, True),
                                                    ("linear", 4, 1, 40, 70, 200, 9, False), ("rbf", 6, 0, 6, 6, 300, 100, False),
                                                    ("rbf", 8, 1, 5, 9, 200, 64, False), ("rbf", 5, 1, 7, 4, 140, 20, True), ("rbf", 7, 0, 4, 5, 129, 128, False),
                                                    ("rbf", 8, 0, 30, 40, 150, 24, False), ("rbf", 3, 1, 6, 5, 300, 128, False), ("rbf", 4, 1, 9, 4, 200, 90, True)])
def test_long_first_paths_take_the_swapped_adjoint(kind, D, d, A, B, M, N, naive, monkeypatch):
    """Gradients of a Gram block whose first paths are long and whose second paths fit the one-band adjoints (route FUSED_SWAP: the
    sweep runs on (y, x), the gradient comes from its second-argument sums; sigkernel.py:404-502 has no such asymmetry) against the
    default routes without the swap (routes.no_adjoint_swap) and the oracle's closed form; values unchanged; mmd through the same route."""
    gen = torch.Generator().manual_seed(D * 100 + M)
    k = sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d, _naive_solver=naive)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    X, Y = Xc.to(DEV), Yc.to(DEV)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    be = _lib.get_backend()
    assert be.route(_lib.OP_ADJOINT, 1 if kind == "rbf" else 0, D, M, N, d, naive, 8) == _lib.ROUTE_FUSED_SWAP
    out = {}
    for off in (False, True):
        monkeypatch.setattr(sigkernel_amd.routes, "no_adjoint_swap", off)
        skmod._route_query.cache_clear()
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        (g1,) = torch.autograd.grad((K * w.to(DEV)).sum(), Xg, retain_graph=True)
        (g2,) = torch.autograd.grad((K * w.to(DEV)).sum(), Xg)
        assert torch.equal(g1, g2)
        out[off] = (K.detach().cpu().numpy(), g1.cpu().numpy())
    monkeypatch.setattr(sigkernel_amd.routes, "no_adjoint_swap", False)
    skmod._route_query.cache_clear()
    assert rel_err(out[False][0], out[True][0]) <= 1e-12 and rel_err(out[False][1], out[True][1]) <= 1e-9
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, d, naive=naive, nthreads=NT)
    assert rel_err(out[False][1], want) <= 1e-9
    assert rel_err(out[False][0], O.gram_forward(Xc, Yc, k, d, naive=naive, nthreads=NT)) <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,M,N", [("linear", 6, 1, 300, 64), ("rbf", 3, 0, 200, 100), ("rbf", 7, 1, 150, 40), ("linear", 2, 2, 140, 33)])
def test_fp32_paths_take_the_swapped_adjoint_up_cast(kind, D, d, M, N, monkeypatch):
    """fp32 paths with long first / short second paths and a gradient: the swapped one-band route on the up-cast paths (as SK_ROUTE_FUSED does
    for fp32 paths: the one-band kernels sweep in fp64 whatever the dtype) -- values and gradients in fp32, within fp32 rounding of the
    oracle on the same fp32 inputs."""
    gen = torch.Generator().manual_seed(7 * D + M)
    k = sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    A, B = 7, 5
    Xc, Yc = walk(gen, A, M, D, torch.float32), walk(gen, B, N, D, torch.float32)
    be = _lib.get_backend()
    assert be.route(_lib.OP_ADJOINT, 1 if kind == "rbf" else 0, D, M, N, d, False, 4) == _lib.ROUTE_FUSED_SWAP
    calls = []
    name = "linear_adjoint_fused" if kind == "linear" else "rbf_adjoint_fused"
    orig = getattr(type(be), name)
    monkeypatch.setattr(type(be), name, lambda self, *a, **kw: (calls.append(kw.get("yside")), orig(self, *a, **kw))[1])
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    Xg = Xc.to(DEV).requires_grad_(True)
    K = sk.compute_Gram(Xg, Yc.to(DEV))
    (K * w.to(DEV).float()).sum().backward()
    assert calls and all(calls), "the swapped adjoint was not used"
    assert K.dtype == torch.float32 and Xg.grad.dtype == torch.float32
    assert rel_err(K.detach().double().cpu().numpy(), O.gram_forward(Xc.double(), Yc.double(), k, d, nthreads=NT)) <= 1e-5
    want = O.gram_grad_weighted(Xc.double(), Yc.double(), w.float().double().numpy(), k, d, nthreads=NT)
    assert rel_err(Xg.grad.double().cpu().numpy(), want) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,A,B,M,N", [("linear", 4, 1, 127, 127, 40, 40), ("rbf", 3, 1, 101, 53, 33, 64), ("linear", 8, 0, 61, 131, 100, 30),
                                              ("rbf", 4, 2, 37, 251, 20, 20), ("rbf", 6, 0, 43, 97, 50, 70), ("linear", 3, 2, 640, 77, 16, 16),
                                              ("rbf", 3, 1, 7, 1009, 24, 24)])
def test_batches_without_a_suitable_divisor_fill_the_lane_groups(kind, D, d, A, B, M, N):
    """The fused adjoints split the B pairs of an x_a into as many chunks as the resident lane groups take; where that number does not
    divide B the chunks differ by one pair (ChunkSplit::uneven, round 6: until then the chunk length had to divide B -- 127 x 127 pairs
    ran 10x slower than 128 x 128).  Gradients against the oracle; the chunks are short (the lane groups are at work)."""
    gen = torch.Generator().manual_seed(A + 3 * B)
    k = sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    be = _lib.get_backend()
    assert be.route(_lib.OP_ADJOINT, 1 if kind == "rbf" else 0, D, M, N, d, False, 8) == _lib.ROUTE_FUSED
    be.last_fused_ppg = None
    Xg = Xc.to(DEV).requires_grad_(True)
    (sk.compute_Gram(Xg, Yc.to(DEV)) * w.to(DEV)).sum().backward()
    assert be.last_fused_ppg is not None and A * -(-B // be.last_fused_ppg) >= min(A * B, 1024), be.last_fused_ppg      # lane groups at work
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, d, nthreads=NT)
    assert rel_err(Xg.grad.cpu().numpy(), want) <= 1e-9
    # the swapped call (the second-argument sums per pair) over the same uneven chunks
    if B > A:
        Yg = Yc.to(DEV).requires_grad_(True)
        (sk.compute_Gram(Yg, Xc.to(DEV)) * w.t().to(DEV)).sum().backward()
        want_y = O.gram_grad_weighted(Yc, Xc, w.t().contiguous().numpy(), k, d, nthreads=NT)
        assert rel_err(Yg.grad.cpu().numpy(), want_y) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,P,M,N", [("linear", 5, 1, 9001, 64, 50), ("linear", 8, 0, 20011, 100, 128), ("linear", 3, 2, 12345, 40, 64),
                                            ("rbf", 3, 1, 9001, 64, 50), ("rbf", 4, 0, 20011, 100, 128), ("rbf", 2, 2, 12345, 40, 64), ("rbf", 3, 1, 5003, 128, 100)])
def test_big_paired_batches_sweep_several_pairs_per_lane_group(kind, D, d, P, M, N, monkeypatch):
    """compute_kernel(X, Y) with a gradient on more pairs than resident lane groups: a lane group of the fused adjoint sweeps several
    consecutive pairs and stores / clears its sums at every pair end (PAIRED; until round 6 one pair per lane group).  Against the
    streaming route on every pair, against the oracle on a sample; one exploding pair in the middle of a lane group's run is rescued."""
    gen = torch.Generator().manual_seed(P)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, d)
    Xc, Yc = walk(gen, P, M, D), walk(gen, P, N, D)
    wild = P // 2 + 3
    reach = 12.0 if kind == "linear" else 20.0
    Xc[wild] = torch.linspace(0, reach, M, dtype=torch.float64)[:, None] * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    Yc[wild] = torch.linspace(0, reach, N, dtype=torch.float64)[:, None] * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    go = torch.randn(P, generator=gen, dtype=torch.float64)
    be = _lib.get_backend()
    res = []
    for off in (False, True):
        monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", off)
        skmod._route_query.cache_clear()
        be.last_fused_ppg = None
        Xg = Xc.to(DEV).requires_grad_(True)
        K = sk.compute_kernel(Xg, Yc.to(DEV))
        (K * go.to(DEV)).sum().backward()
        res.append((K.detach().cpu().numpy(), Xg.grad.cpu().numpy(), be.last_fused_ppg))
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", False)
    skmod._route_query.cache_clear()
    assert res[0][2] is not None and res[0][2] > 1, res[0][2]          # several pairs per lane group
    assert abs(res[0][0][wild]) > 1e5
    assert rel_err(res[0][0], res[1][0]) <= 1e-12
    for p in range(P):      # row by row: the wild pair's gradient is 1e6 times the others'
        # (the exploding pair comes from two different stored-grid rescues of a kernel of 1e8: held to the oracle below)
        if rel_err(res[0][1][p], res[1][1][p]) > (1e-8 if p != wild else 1e-5): raise AssertionError((p, rel_err(res[0][1][p], res[1][1][p])))
    for p in (0, 1, wild - 1, wild, wild + 1, P - 1):
        want = O.gram_grad_weighted(Xc[p:p + 1], Yc[p:p + 1], go[p:p + 1].reshape(1, 1).numpy(), k, d)
        # (the rbf pair of |K| ~ 1e8: two evaluations of its 65 x 51 exponentials differ in the last bit, the kernel amplifies that)
        assert rel_err(res[0][1][p], want[0]) <= (1e-6 if (p == wild and kind == "rbf") else 2 * be.ADJ_RESIDUAL_TOL), p


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rbf", "linear"])
def test_streaming_gradient_reuses_the_increments_its_forward_kept(kind, monkeypatch):
    """A one-tile Gram block on the streaming route with a gradient pending keeps its increments beside the edges (cost entry
    keep_increments_fraction of the budget): backward evaluates the static kernel ONCE less and returns the same bits; with the entry
    at 0, or a budget the increments do not fit, they are formed again."""
    from sigkernel_amd import _lib, sigkernel
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(123)
    X, Y = walk(gen, 6, 40, 20).to(DEV), walk(gen, 5, 50, 20).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel(), 1)
    calls = []
    orig = type(be).static_increments
    monkeypatch.setattr(type(be), "static_increments", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])

    def step():
        del calls[:]
        Xg = X.clone().requires_grad_(True)
        sk.compute_Gram(Xg, Y).sum().backward()
        return Xg.grad, len(calls)
    g_kept, n_kept = step()
    monkeypatch.setattr(sigkernel, "_KEEP_INCREMENTS_FRACTION", 0.0)
    g_again, n_again = step()
    assert (n_kept, n_again) == (1, 2) and torch.equal(g_kept, g_again)
    monkeypatch.setattr(sigkernel, "_KEEP_INCREMENTS_FRACTION", None)
    sk_small = sigkernel_amd.SigKernel(sk.static_kernel, 1, workspace_bytes=6 * 5 * 39 * 64 * 8 * 4)      # (the increments: more than an eighth of it)
    Xg = X.clone().requires_grad_(True)
    del calls[:]
    sk_small.compute_Gram(Xg, Y).sum().backward()
    assert len(calls) >= 2 and torch.allclose(Xg.grad, g_kept, rtol=1e-12, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d", [("rbf", 20, 1), ("linear", 12, 1), ("rbf", 3, 3)])
def test_symmetric_forward_on_the_streaming_route_takes_the_blocked_triangle(kind, D, d, monkeypatch):
    """compute_Gram(X, X, sym=True) without a gradient on the streaming route (wide paths, dyadic 3): from sym_stream_min_paths paths on
    only the blocks on and above the diagonal are solved (the node evaluation and the increments are most of a pair's cost there) --
    exactly symmetric, equal to the full computation."""
    gen = torch.Generator().manual_seed(D + d)
    k = sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    A = int(_lib.cost("sym_stream_min_paths")) + 3
    X = walk(gen, A, 24, D).to(DEV)
    from sigkernel_amd import sigkernel as S
    calls = []
    orig = S._gram_block
    monkeypatch.setattr(S, "_gram_block", lambda *a, **kw: (calls.append(a[2].shape[0]), orig(*a, **kw))[1])
    K = sk.compute_Gram(X, X, sym=True)
    assert len(calls) == int(_lib.cost("sym_tiles")) and sum(calls) == A, calls        # row blocks, not one block
    monkeypatch.setattr(S, "_gram_block", orig)
    K2 = sk.compute_Gram(X, X, sym=False)
    assert torch.equal(K, K.t()) and rel_err(K.cpu().numpy(), K2.cpu().numpy()) <= 1e-12
    # fewer paths: one block of all pairs
    calls.clear()
    monkeypatch.setattr(S, "_gram_block", lambda *a, **kw: (calls.append(a[2].shape[0]), orig(*a, **kw))[1])
    sk.compute_Gram(X[:100], X[:100], sym=True)
    assert calls == [100], calls


def _mb_split_knob(on):
    os.environ["SK_FUSEDMB_SPLIT"] = "1" if on else "0"
    _lib.load().sk_reload_knobs()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,A,B,M,N,dt", [("rbf", 3, 1, 2, 2, 2048, 2048, torch.float64), ("linear", 8, 1, 3, 2, 700, 1100, torch.float64),
                                                 ("rbf", 4, 0, 1, 5, 1030, 1024, torch.float64), ("linear", 4, 0, 2, 3, 1500, 515, torch.float64),
                                                 ("rbf", 2, 2, 1, 1, 300, 600, torch.float64), ("rbf", 16, 2, 2, 3, 200, 512, torch.float32),
                                                 ("linear", 12, 2, 1, 2, 129, 2000, torch.float64), ("rbf", 9, 1, 4, 1, 400, 544, torch.float64)])
def test_bands_of_a_pair_on_several_waves(kind, D, d, A, B, M, N, dt):
    """Few pairs of long paths (csrc/sk_wave_fused_mb.hip, split mode): the bands of a pair on different waves, trailing each other
    through the pair's boundary rows and a progress counter -- bit-identical to the one-wave-per-pair sweep (SK_FUSEDMB_SPLIT=0),
    bitwise reproducible, and equal to the oracle (cython_backend.pyx:64-119: no length limit); second paths of 2^k points (N - 1 = 15
    mod 16: the padding unit behind the last real one) included."""
    gen = torch.Generator().manual_seed(M + N + D)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.8)
    sk = sigkernel_amd.SigKernel(k, d)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    X, Y = Xc.to(dt).to(DEV), Yc.to(dt).to(DEV)
    lib = _lib.load()
    try:
        _mb_split_knob(True)
        assert lib.sk_solve_fwd_static_split(0 if kind == "linear" else 1, A * B, M - 1, N - 1, d, D) >= 2      # the mode under test is the one taken
        K = sk.compute_Gram(X, Y)
        # the launch's status word (the last 8 bytes of its workspace): no item's bounded wait for the band above gave up
        st = _lib.get_backend().last_split_status
        assert st is None or int(st) == 0      # (None: the route swapped the arguments and that orientation ran one wave per pair, or fp32 staging)
        K2 = sk.compute_Gram(X, Y)
        Kp = sk.compute_kernel(X[:1], Y[:1])
        _mb_split_knob(False)
        assert lib.sk_solve_fwd_static_split(0 if kind == "linear" else 1, A * B, M - 1, N - 1, d, D) == 0
        K1 = sk.compute_Gram(X, Y)
    finally:
        os.environ.pop("SK_FUSEDMB_SPLIT", None)
        lib.sk_reload_knobs()
    assert torch.equal(K, K1) and torch.equal(K, K2) and torch.equal(Kp[0], K[0, 0])
    want = O.gram_forward(Xc.to(dt), Yc.to(dt), k, d, nthreads=NT)
    assert rel_err(K.double().cpu().numpy(), want) <= (1e-4 if dt == torch.float32 else 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,M,N", [("linear", 8, 1, 40, 33), ("rbf", 3, 2, 30, 30), ("rbf", 6, 0, 300, 260), ("linear", 20, 1, 25, 25)])
def test_pair_limit_of_a_launch_tiles_over_rows(kind, D, d, M, N, monkeypatch):
    """More pairs than one fused launch indexes (_MAX_LAUNCH_PAIRS): row tiles, forward with kept edges and backward alike -- the same
    bits as the untiled call on every route (one band, multi-band, streamed)."""
    gen = torch.Generator().manual_seed(8)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.9)
    sk = sigkernel_amd.SigKernel(k, d)
    X, Y = walk(gen, 23, M, D).to(DEV), walk(gen, 9, N, D).to(DEV)
    w = torch.randn(23, 9, generator=gen, dtype=torch.float64).to(DEV)
    out = []
    for limit in (1 << 30, 4 * 9 + 1):
        monkeypatch.setattr(skmod, "_MAX_LAUNCH_PAIRS", limit)
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        (K * w).sum().backward()
        out.append((K.detach(), Xg.grad, sk.compute_Gram(X, Y)))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2])
    assert rel_err(out[1][1].cpu().numpy(), out[0][1].cpu().numpy()) <= 1e-12      # (partial sums of a row are added per tile)


def _second_stream_cases(gen):
    lin, rbf = sigkernel_amd.LinearKernel(), sigkernel_amd.RBFKernel(1.0)

    def gram(sk, X, Y):
        return lambda: sk.compute_Gram(X, Y)

    def gram_bwd(sk, X, Y, w):
        def f():
            Xg = X.clone().requires_grad_(True)
            (sk.compute_Gram(Xg, Y) * w).sum().backward()
            return Xg.grad
        return f
    X, Y = walk(gen, 512, 128, 8).to(DEV), walk(gen, 512, 128, 8).to(DEV)
    w = torch.randn(512, 512, generator=gen, dtype=torch.float64).to(DEV)
    X4, Y4 = walk(gen, 512, 64, 4).to(DEV), walk(gen, 512, 64, 4).to(DEV)
    X20, Y20 = walk(gen, 256, 128, 20).to(DEV), walk(gen, 256, 128, 20).to(DEV)
    Xd, Yd, gam = (walk(gen, 256, 128, 4).to(DEV) for _ in range(3))
    return {"fused_forward": gram(sigkernel_amd.SigKernel(lin, 1), X, Y),                                       # F1: work queue
            "fused_linear_adjoint": gram_bwd(sigkernel_amd.SigKernel(lin, 1), X, Y, w),                       # A1: chunks by age rank
            "fused_rbf_adjoint": gram_bwd(sigkernel_amd.SigKernel(rbf, 2), X4, Y4, w),                        # A1
            "streaming_forward_and_adjoint": gram_bwd(sigkernel_amd.SigKernel(lin, 1), X20, Y20, w[:256, :256].contiguous()),   # S1, S2: shares by age rank
            "derivative_solver": lambda: torch.stack(sigkernel_amd.SigKernel(lin, 1).compute_kernel_and_derivatives_Gram(Xd, Yd, gam))}   # D2


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["fused_forward", "fused_linear_adjoint", "fused_rbf_adjoint", "streaming_forward_and_adjoint", "derivative_solver"])
def test_second_stream_busy_same_bits_little_slowdown(family):
    """Work someone else has on the chip must cost neither bits nor much time, whichever way a family hands out its pairs -- the fused
    forwards from a per-launch work queue, the streaming solver / adjoint, the derivative solver and the one-band fused adjoints as
    shares by the wave's age rank on its SIMD (blockIdx / #CU, sk_wave_common.h).  A side stream is kept busy for the WHOLE timed
    region by a hipGraph of 2000 small launches (a Python loop of launches drains as fast as it is enqueued): every family is
    bit-identical and < 15 % slower (measured 2-6 %, the age-rank families no more than the queue ones; against a tenant that wants
    the whole chip all of them share it alike: profiles/r04_second_stream.txt)."""
    import time
    f = _second_stream_cases(torch.Generator().manual_seed(5))[family]
    side = torch.cuda.Stream()
    buf = torch.zeros(1 << 16, dtype=torch.float32, device=DEV)      # 256 KB: each launch occupies a few CUs for microseconds
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            buf.add_(1.0)
        noise = torch.cuda.CUDAGraph()
        with torch.cuda.graph(noise, stream=side):
            for _ in range(2000):
                buf.add_(1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        noise.replay()
        side.synchronize()
        noise_s = time.perf_counter() - t0

    def timed(reps, load, stream=side):
        torch.cuda.synchronize()
        if load:
            with torch.cuda.stream(stream):
                for _ in range(int(2.5 * load / noise_s) + 2):
                    noise.replay()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.current_stream().synchronize()
        dt = (time.perf_counter() - t0) / reps
        still = not stream.query()
        stream.synchronize()
        return dt, r, still
    for _ in range(5):
        f()
    one = timed(3, 0)[0]
    reps = max(3, min(20, int(0.12 / one)))
    alone, r0, _ = min((timed(reps, 0) for _ in range(3)), key=lambda r: r[0])
    # HIP multiplexes streams onto a few hardware queues: a side stream that lands on the queue of the caller's stream is served IN
    # ORDER with it (measured: +25 % then, whichever family) -- that is the runtime's queue, not a second tenant.  Three consecutive
    # streams of torch's pool cannot all share it: the least disturbed of them is the measurement.
    best = None
    for stream in (side, torch.cuda.Stream(), torch.cuda.Stream()):
        load = alone * reps
        for attempt in range(3):          # (a side stream that ran dry inside the timed region measured nothing: load it more)
            res = [timed(reps, load, stream) for _ in range(2)]
            if all(r[2] for r in res):
                break
            load *= 2.5
        assert all(torch.equal(r0, r[1]) for r in res)
        assert all(r[2] for r in res), "the side stream ran dry inside the timed region three times over"
        busy = min(r[0] for r in res)
        best = busy if best is None else min(best, busy)
        if best <= 1.08 * alone:
            break
    assert best <= 1.15 * alone, (family, alone, best)      # measured +2 .. +6 % over several boxes


def _one_wild_pair(gen, A, B, M, D):
    """Random walks, except that x_2 and y_5 are the same straight line: k(x_2, y_5) explodes (1e6 .. 1e15), every other pair is tame."""
    X, Y = walk(gen, A, M, D) * 2, walk(gen, B, M, D) * 2
    line = torch.arange(M, dtype=torch.float64)[:, None] * 0.6 * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    X[2], Y[5] = line, line.clone()
    return X, Y


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d", [("linear", 4, 1), ("rbf", 4, 2), ("rbf", 3, 1)])
@pytest.mark.parametrize("screen", [1e3, 1e300])
def test_fused_adjoints_rescue_an_exploding_pair_on_the_device(kind, D, d, screen, monkeypatch):
    """Failure injection: one pair of a Gram block has |K| ~ 1e6 .. 1e15 -- far beyond what the fused adjoints' backward
    recompute of K survives.  With the default screen the pair is taken out of the sweep (residual entry -1) and its exact,
    stored-grid share is added on the device; with the screen disabled (1e300) it fails its self-check after the fact and its
    whole chunk is recomputed exactly.  Either way every row of the gradient matches the oracle, and nothing is read back."""
    be = _lib.get_backend()
    monkeypatch.setattr(type(be), "FUSED_SCREEN", screen)
    gen = torch.Generator().manual_seed(41)
    Xc, Yc = _one_wild_pair(gen, 6, 40, 32, D)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    wc = torch.randn(6, 40, generator=gen, dtype=torch.float64)
    Kc = O.gram_forward(Xc, Yc, k, d)
    wild = np.abs(Kc) > 1e3
    assert np.abs(Kc[2, 5]) > 1e5 and 1 <= wild.sum() <= 12
    be.last_fused_err = None
    Xg = Xc.to(DEV).requires_grad_(True)
    calls = []
    name = "linear_adjoint_fused" if kind == "linear" else "rbf_adjoint_fused"
    orig = getattr(type(be), name)
    monkeypatch.setattr(type(be), name, lambda self, *a, **kw: (calls.append(kw.get("kfinal") is not None), orig(self, *a, **kw))[1])
    (sigkernel_amd.SigKernel(k, d).compute_Gram(Xg, Yc.to(DEV)) * wc.to(DEV)).sum().backward()
    assert calls and all(calls), "the fused adjoint was not used, or not armed with the forward values"
    assert be.last_fused_err is not None, "the fused adjoint declined the case"
    err = be.last_fused_err.cpu().numpy().reshape(6, 40)
    if screen < 1e100:
        assert np.array_equal(err < 0, wild) and np.all(err[wild] == -1.0)
    else:
        assert err[2, 5] > be.ADJ_RESIDUAL_TOL or np.isnan(err[2, 5])
    want = O.gram_grad_weighted(Xc, Yc, wc.numpy(), k, d, nthreads=NT)
    got = Xg.grad.cpu().numpy()
    # row by row (the wild pair's row is 1e6 times larger than the others), to twice the self-check bound the fused sweep accepts
    # for the pairs it keeps: their backward recompute of K loses ~1e-16 K^2, and K reaches 8e2 below the screen here
    for a in range(6):
        assert rel_err(got[a], want[a]) <= 2 * be.ADJ_RESIDUAL_TOL, (a, rel_err(got[a], want[a]))


@pytest.mark.gpu
def test_fused_rescue_covers_the_second_argument_sums(monkeypatch):
    """compute_Gram(X, X, sym=True) in triangular row blocks with a gradient and ONE exploding pair above the diagonal: the rescue
    must also deliver that pair's second-argument sums (its mirror image's share of the gradient)."""
    from sigkernel_amd import sigkernel as S
    monkeypatch.setattr(S, "_SYM_TILES", 3)
    monkeypatch.setattr(S, "_SYM_MIN_CELLS", 0.0)
    monkeypatch.setattr(S, "_SYM_MIN_ROWS", 4)
    gen = torch.Generator().manual_seed(43)
    Xc = walk(gen, 30, 33, 4) * 2
    line = torch.arange(33, dtype=torch.float64)[:, None] * 0.6 * torch.ones(1, 4, dtype=torch.float64) / 2.0
    Xc[3], Xc[17] = line, line + 0.01           # (3, 17): rows in different blocks -> reaches row 17 through the second argument
    k = sigkernel_amd.RBFKernel(1.0)
    w = torch.randn(30, 30, generator=gen, dtype=torch.float64)
    Xg = Xc.to(DEV).requires_grad_(True)
    (sigkernel_amd.SigKernel(k, 2).compute_Gram(Xg, Xg, sym=True) * w.to(DEV)).sum().backward()
    want = 2.0 * O.gram_grad_weighted(Xc, Xc, w.numpy(), k, 2, nthreads=NT)       # the reference's 2x rule
    got = Xg.grad.cpu().numpy()
    for a in range(30):
        assert rel_err(got[a], want[a]) <= 2 * _lib.HipBackend.ADJ_RESIDUAL_TOL, (a, rel_err(got[a], want[a]))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,d,N", [("linear", 4, 1, 32), ("linear", 4, 0, 80), ("linear", 4, 2, 20), ("rbf", 6, 0, 80), ("rbf", 6, 1, 32)])
@pytest.mark.parametrize("screen", [1e3, 1e300])
def test_swapped_adjoint_rescues_an_exploding_pair(kind, D, d, N, screen, monkeypatch):
    """Long first paths against short second ones (route FUSED_SWAP on (y, x) with the second-argument sums: sk_linear_adjoint_fused_f64,
    and sk_rbf_adjoint_fused_f64 on paths of dim 5..8, where the sums replace the first-argument ones), one pair with |K| ~ 1e5 .. 1e9:
    screened out of the sweep -- or, with the screen disabled, failing its self-check after the fact -- its block of the sums comes
    from the stored-grid rescue; every row of the gradient matches the oracle."""
    be = _lib.get_backend()
    monkeypatch.setattr(type(be), "FUSED_SCREEN", screen)
    gen = torch.Generator().manual_seed(47 + d)
    A, B, M = 6, 40, 200
    Xc, Yc = walk(gen, A, M, D) * 2, walk(gen, B, N, D) * 2
    reach = 12.0 if kind == "linear" else 20.0
    Xc[2] = torch.linspace(0, reach, M, dtype=torch.float64)[:, None] * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    Yc[5] = torch.linspace(0, reach, N, dtype=torch.float64)[:, None] * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    assert be.route(_lib.OP_ADJOINT, 0 if kind == "linear" else 1, D, M, N, d, False, 8) == _lib.ROUTE_FUSED_SWAP
    wc = torch.randn(A, B, generator=gen, dtype=torch.float64)
    Kc = O.gram_forward(Xc, Yc, k, d)
    wild = np.abs(Kc) > 1e3
    assert np.abs(Kc[2, 5]) > 1e5 and 1 <= wild.sum() <= 12
    be.last_fused_err = None
    calls = []
    name = "linear_adjoint_fused" if kind == "linear" else "rbf_adjoint_fused"
    orig = getattr(type(be), name)
    monkeypatch.setattr(type(be), name, lambda self, *a, **kw: (calls.append((kw.get("kfinal") is not None, kw.get("yside"))), orig(self, *a, **kw))[1])
    Xg = Xc.to(DEV).requires_grad_(True)
    (sigkernel_amd.SigKernel(k, d).compute_Gram(Xg, Yc.to(DEV)) * wc.to(DEV)).sum().backward()
    assert calls and all(c == (True, True) for c in calls), "the swapped adjoint was not used, or not armed with the forward values"
    err = be.last_fused_err.cpu().numpy().reshape(B, A)          # pairs (b, a)
    if screen < 1e100:
        assert np.array_equal(err < 0, wild.T) and np.all(err[wild.T] == -1.0)
    else:
        assert err[5, 2] > be.ADJ_RESIDUAL_TOL or np.isnan(err[5, 2])
    want = O.gram_grad_weighted(Xc, Yc, wc.numpy(), k, d, nthreads=NT)
    got = Xg.grad.cpu().numpy()
    for a in range(A):
        assert rel_err(got[a], want[a]) <= 2 * be.ADJ_RESIDUAL_TOL, (a, rel_err(got[a], want[a]))


@pytest.mark.gpu
def test_one_exploding_pair_costs_one_stored_grid_solve_not_a_recompute():
    """1 wild pair among 1 048 576 (512 x 2048 pairs of the BASELINE configs[3] shape): the backward pass must stay within 1.1x
    of the same pass without it (round 2 threw the whole fused gradient away and recomputed everything unfused: ~2x)."""
    gen = torch.Generator().manual_seed(44)
    A, B, M, D = 512, 2048, 64, 4
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, M, D)
    line = torch.arange(M, dtype=torch.float64)[:, None] * 0.6 * torch.ones(1, D, dtype=torch.float64) / 2.0
    Xw, Yw = Xc.clone(), Yc.clone()
    Xw[2], Yw[5] = line, line.clone()
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64).to(DEV)

    def bwd_ms(X, Y):
        X, Y = X.to(DEV), Y.to(DEV)
        best = 1e9
        for _ in range(4):
            Xg = X.clone().requires_grad_(True)
            loss = (sk.compute_Gram(Xg, Y) * w).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            loss.backward()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best, Xg.grad
    t_tame, _ = bwd_ms(Xc, Yc)
    t_wild, g = bwd_ms(Xw, Yw)
    assert (_lib.get_backend().last_fused_err < 0).sum().item() >= 1
    assert bool(torch.isfinite(g).all())
    assert t_wild <= 1.1 * t_tame + 0.5, (t_tame, t_wild)


@pytest.mark.gpu
def test_adjoint_workspace_is_the_fast_one_for_long_paths():
    """solve_adj without kept edges used to allocate min(P, 1024) whole pairs of solution grids (68 GB for 1024
    pairs of 2044 x 2044 grids).  The fast route needs P (edges + 1) doubles; the rescue scratch is capped."""
    lib = _lib.load()
    P, Mc, Nc, d = 1024, 511, 511, 2
    fast = int(lib.sk_adj_workspace_bytes(P, Mc, Nc, d, _lib.FLAG_FAST_ONLY, 8))
    full = int(lib.sk_adj_workspace_bytes(P, Mc, Nc, d, 0, 8))
    assert 0 < fast < 64 << 20 and full > 60 << 30
    be = _lib.HipBackend()
    ws, nbytes = be._grid_scratch(P, Mc, Nc, d, torch.device(DEV), be.RESCUE_SLOTS)
    assert nbytes <= be.GRID_SCRATCH_BYTES and nbytes >= int(lib.sk_adj_rescue_slot_bytes(Mc, Nc, d))
    # and it works end to end on a few such pairs, fp64 and fp32 (the latter through the up-cast chunks)
    rng = np.random.default_rng(1)
    inc = rng.normal(scale=0.004, size=(3, 511, 511))
    want_k, want_w = O.adjoint_coarse(inc, 2, nthreads=NT)
    ldp = _lib._padded_ld(511, 8)
    buf = torch.zeros(3, 511, ldp, dtype=torch.float64, device=DEV)
    buf[..., :511] = torch.from_numpy(inc).to(DEV)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    k, W = be.solve_adj(buf[..., :511], 2)
    assert torch.cuda.max_memory_allocated() - base < 3 << 30
    assert rel_err(W.cpu().numpy(), want_w) <= 1e-9 and rel_err(k.cpu().numpy(), want_k) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["rbf", "linear"])
def test_nan_coordinates_give_nan_on_every_route(kernel, monkeypatch):
    """A NaN or infinite coordinate must poison the kernel value on the fused AND the unfused forward route, like the
    reference's exp of a NaN exponent does -- a diverged generator must not see a finite loss."""
    gen = torch.Generator().manual_seed(2)
    X, Y = walk(gen, 4, 30, 3).to(DEV), walk(gen, 3, 25, 3).to(DEV)
    X[1, 7, 2] = float("nan")
    X[2, 3, 0] = float("inf")
    k = sigkernel_amd.RBFKernel(0.7) if kernel == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, 1)
    for env in ("", "1"):
        if env:
            monkeypatch.setattr(sigkernel_amd.routes, "no_fused_rbf", True)
        K = sk.compute_Gram(X, Y)
        assert torch.isnan(K[1]).all(), (kernel, env, K)
        assert torch.isfinite(K[0]).all() and torch.isfinite(K[3]).all()
        # an infinite coordinate: the reference's |x|^2 + |y|^2 - 2<x,y> is inf - inf = NaN (static_kernels.py:70-73), linear
        # increments are inf - inf as well: the row is poisoned on every route
        assert not torch.isfinite(K[2]).any(), (kernel, env, K)


# ---------------------------------------------------------------------------------------------
# the multi-band fused forward (sk_solve_fwd_static_*, csrc/sk_wave_fused_mb.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fused_multiband_forward_against_the_oracle_and_the_streaming_route():
    """LinearKernel / RBFKernel formed inside the solver for pairs that need several bands of a wavefront and path dims up to
    16: 90 random shapes (both kinds, dyadic 0..2, dims 1..16, 1..10 bands, ragged second paths, paired and Gram, many pairs
    per wave, fp32 and fp64 tensors -- fp32 RBF at dim > 8 takes the packed-fp32 ring) against sk_static_increments +
    sk_solve_fwd (1e-11 / fp32: 2e-6), every fifth one also against the CPU oracle."""
    from sigkernel_amd.sigkernel import _increments
    be = _lib.get_backend()
    rng = np.random.default_rng(0)
    n = 0
    for it in range(90):
        kind = int(rng.integers(0, 2))
        d = int(rng.integers(0, 3))
        D = int(rng.integers(1, 17))
        M = int(rng.integers(2, 700 >> d)) if it % 3 else int(rng.integers(60, 140))
        N = int(rng.integers(160, 420))
        A, B = int(rng.integers(1, 6)), int(rng.integers(1, 8))
        gram = bool(it % 4)
        if it % 7 == 6:
            A, B, M, N = 40, 90, int(rng.integers(20, 200 >> d) + 2), 160 + int(rng.integers(0, 30))    # many pairs per wave
        if not gram:
            B = A
        dt = torch.float32 if it % 5 == 4 else torch.float64
        gen = torch.Generator().manual_seed(100 + it)
        Xc, Yc = walk(gen, A, M, D, dt) * 1.5, walk(gen, B, N, D, dt) * 1.5
        X, Y = Xc.to(DEV), Yc.to(DEV)
        sk = sigkernel_amd.LinearKernel(0.9) if kind == 0 else sigkernel_amd.RBFKernel(0.8)
        par = (1.0 if gram else 0.9) if kind == 0 else 0.8
        K = be.solve_fwd_fused_static(kind, par, X, Y, d, False, gram)
        assert K is not None and K.dtype == dt, (it, kind, d, D, M, N)
        want = be.solve_fwd(_increments(be, sk, X.double(), Y.double(), gram), d)
        tol = 1e-11 if dt == torch.float64 else 2e-6
        assert rel_err(K.double().cpu().numpy(), want.cpu().numpy()) <= tol, (it, kind, d, D, A, B, M, N, gram, dt)
        if it % 5 == 0 and gram and A * B * (M << d) * (N << d) < 3e8:
            ref = O.gram_forward(Xc.double(), Yc.double(), sk, d, nthreads=NT)
            assert rel_err(K.double().cpu().numpy(), ref) <= tol
        n += 1
    assert n == 90


@pytest.mark.gpu
def test_fused_multiband_scope_and_c5_route(monkeypatch):
    """Outside its scope the kernel says so (the caller falls back to increments in HBM); inside it -- the C5 shape -- it is
    what compute_Gram runs: nothing of size pairs x M x N is materialised (sk_static_increments is never called)."""
    be = _lib.get_backend()
    Z = lambda A, M, D, dt=torch.float64: torch.zeros(A, M, D, dtype=dt, device=DEV)
    # (round 4: short second paths are swept with padding units -- no shape of dim <= 16, dyadic <= 2 is outside the scope)
    assert torch.equal(be.solve_fwd_fused_static(1, 1.0, Z(2, 100, 3), Z(2, 100, 3), 1, False, True), torch.ones(2, 2, dtype=torch.float64, device=DEV))
    # a long first path against a short second one: either orientation (the kernel is symmetric; sk_route_query picks the cheaper)
    gen0 = torch.Generator().manual_seed(4)
    Xs, Ys = walk(gen0, 3, 400, 5).to(DEV), walk(gen0, 4, 90, 5).to(DEV)
    for kind, kern in ((0, sigkernel_amd.LinearKernel()), (1, sigkernel_amd.RBFKernel(0.9))):
        for swap in (False, True):
            Ksw = be.solve_fwd_fused_static(kind, 1.0 if kind == 0 else 0.9, Xs, Ys, 1, False, True, swap=swap)
            assert Ksw is not None and Ksw.shape == (3, 4)
            assert rel_err(Ksw.cpu().numpy(), O.gram_forward(Xs.cpu(), Ys.cpu(), kern, 1, nthreads=NT)) <= 1e-11
    Kp = be.solve_fwd_fused_static(0, 0.8, Xs, Ys[:3].contiguous(), 1, False, False)
    Gp = sigkernel_amd.LinearKernel(0.8).batch_kernel(Xs.cpu(), Ys[:3].cpu()).numpy()
    assert rel_err(Kp.cpu().numpy(), O.solve_coarse(O.increments(Gp), 1)) <= 1e-11
    assert be.solve_fwd_fused_static(1, 1.0, Z(2, 300, 17), Z(2, 300, 17), 1, False, True) is None     # dim 17
    assert be.solve_fwd_fused_static(1, 1.0, Z(2, 300, 3), Z(2, 300, 3), 3, False, True) is None       # dyadic 3
    Kn = be.solve_fwd_fused_static(0, 1.0, Xs, Ys, 1, True, True)                                        # naive scheme: a launch-time constant
    assert rel_err(Kn.cpu().numpy(), O.gram_forward(Xs.cpu(), Ys.cpu(), sigkernel_amd.LinearKernel(), 1, naive=True, nthreads=NT)) <= 1e-11
    gen = torch.Generator().manual_seed(8)
    X, Y = walk(gen, 3, 512, 16, torch.float32).to(DEV), walk(gen, 5, 512, 16, torch.float32).to(DEV)
    sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 2)
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_mb", True)
    K_stream = sk.compute_Gram(X, Y)
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_mb", False)
    monkeypatch.setattr(type(be), "static_increments", lambda self, *a, **k: (_ for _ in ()).throw(AssertionError("increments materialised")))
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    K = sk.compute_Gram(X, Y)
    assert torch.cuda.max_memory_allocated() - base < 600 << 20      # workspace rows only (the increments would be 16 MB per pair)
    assert rel_err(K.cpu().numpy(), K_stream.cpu().numpy()) <= 2e-6
    # fp64 tensors of the same shape: the fp64 ring (16 staged fp64 dims: by default such calls stream -- measured faster --, the
    # memory-first switch keeps them fused)
    monkeypatch.setattr(sigkernel_amd.routes, "no_stream", True)
    K64 = sk.compute_Gram(X.double(), Y.double())
    assert rel_err(K64.cpu().numpy(), O.gram_forward(X.double().cpu(), Y.double().cpu(), sigkernel_amd.RBFKernel(1.0), 2, nthreads=NT)) <= 1e-11
    # LinearKernel on long paths (two bands at dyadic 1) goes the same way
    Xl, Yl = walk(gen, 4, 300, 8).to(DEV), walk(gen, 3, 280, 8).to(DEV)
    Kl = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1).compute_Gram(Xl, Yl)
    assert rel_err(Kl.cpu().numpy(), O.gram_forward(Xl.cpu(), Yl.cpu(), sigkernel_amd.LinearKernel(), 1, nthreads=NT)) <= 1e-11


# ---------------------------------------------------------------------------------------------
# the multi-band fused RBF adjoint (sk_rbf_adjoint_fused_mb_f64, csrc/sk_wave_adj_fused_mb.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,A,B,M,N,D", [(2, 3, 4, 150, 170, 3), (2, 2, 3, 60, 200, 10), (1, 3, 2, 300, 180, 5), (2, 2, 2, 129, 161, 16),
                                         (1, 2, 2, 130, 300, 12), (2, 5, 7, 64, 165, 4), (2, 2, 3, 65, 162, 8), (1, 2, 2, 257, 161, 2)])
def test_multiband_fused_rbf_adjoint_vs_oracle(d, A, B, M, N, D):
    """Long / wide paths: terminal edges from sk_solve_fwd_static_* and the multi-band fused adjoint against the oracle's closed
    form (O.gram_grad_weighted), row by row -- band boundaries (M = 64 k + 1, 129, 257), one band, 1..3 bands, dims up to 16,
    N at the edge of a unit row (odd / even Nc)."""
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(M * 7 + N)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    k = sigkernel_amd.RBFKernel(0.7)
    res = be.solve_fwd_fused_static(1, 0.7, Xc.to(DEV), Yc.to(DEV), d, False, True, keep_edges=True)
    assert res is not None and res[1] is not None
    K, edges = res
    assert rel_err(K.cpu().numpy(), O.gram_forward(Xc, Yc, k, d, nthreads=NT)) <= 1e-11
    out = be.rbf_adjoint_fused_mb(Xc.to(DEV), Yc.to(DEV), 0.7, d, edges, w.reshape(-1).to(DEV), gram=True)
    assert out is not None
    got, resid = out[0].cpu().numpy(), float(out[1])
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, d, nthreads=NT)
    assert resid <= _lib.HipBackend.ADJ_RESIDUAL_TOL
    scale = np.abs(want).max()
    for r in range(M):
        assert np.abs(got[:, r] - want[:, r]).max() <= 1e-10 * scale, (r, np.abs(got[:, r] - want[:, r]).max() / scale)
    # paired batch through the same kernel
    n = min(A, B)
    resp = be.solve_fwd_fused_static(1, 0.7, Xc[:n].to(DEV), Yc[:n].to(DEV), d, False, False, keep_edges=True)
    outp = be.rbf_adjoint_fused_mb(Xc[:n].to(DEV), Yc[:n].to(DEV), 0.7, d, resp[1], w.diagonal()[:n].contiguous().to(DEV), gram=False)
    for i in range(n):
        wi = O.gram_grad_weighted(Xc[i:i + 1], Yc[i:i + 1], w[i:i + 1, i:i + 1].numpy(), k, d)[0]
        assert np.abs(outp[0][i].cpu().numpy() - wi).max() <= 1e-10 * max(np.abs(wi).max(), 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("d,A,B,M,N,D", [(1, 3, 4, 300, 170, 3), (0, 2, 3, 300, 200, 10), (2, 3, 2, 150, 180, 5), (1, 2, 2, 129, 161, 16),
                                         (0, 2, 2, 257, 300, 12), (2, 5, 7, 64, 165, 4), (1, 2, 3, 140, 161, 8), (0, 3, 2, 513, 160, 2)])
def test_multiband_fused_linear_adjoint_vs_oracle(d, A, B, M, N, D, monkeypatch):
    """LinearKernel on long / wide paths: edges from sk_solve_fwd_static_* (kind 0) + sk_linear_adjoint_fused_mb_f64 against the oracle's
    closed form, row by row (1..3 bands, band boundaries, dims up to 16, dyadic 0..2, LinearKernel(scale)); through the API no
    increments and no W are formed (sk_static_increments / sk_solve_adj never called)."""
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(M * 5 + N)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    k = sigkernel_amd.LinearKernel()
    for name in ("static_increments", "solve_adj", "static_adjoint"):
        monkeypatch.setattr(type(be), name, (lambda nm: (lambda self, *a, **kw: (_ for _ in ()).throw(AssertionError(nm + " called"))))(name))
    Xg = Xc.to(DEV).requires_grad_(True)
    K = sigkernel_amd.SigKernel(k, d).compute_Gram(Xg, Yc.to(DEV))
    assert rel_err(K.detach().cpu().numpy(), O.gram_forward(Xc, Yc, k, d, nthreads=NT)) <= 1e-11
    (K * w.to(DEV)).sum().backward()
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, d, nthreads=NT)
    got = Xg.grad.cpu().numpy()
    scale = np.abs(want).max()
    for r in range(M):
        assert np.abs(got[:, r] - want[:, r]).max() <= 1e-10 * scale, (r, np.abs(got[:, r] - want[:, r]).max() / scale)
    assert float(be.last_fused_err.max()) <= _lib.HipBackend.ADJ_RESIDUAL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind,d,M,N,D,dt", [("rbf", 1, 300, 200, 12, torch.float64), ("rbf", 2, 150, 200, 12, torch.float64),
                                             ("rbf", 2, 150, 200, 16, torch.float32), ("rbf", 1, 300, 200, 16, torch.float32),
                                             ("linear", 1, 300, 200, 12, torch.float64), ("linear", 2, 150, 200, 12, torch.float64),
                                             ("linear", 2, 150, 200, 16, torch.float32), ("linear", 0, 300, 200, 12, torch.float64),
                                             ("linear", 2, 150, 200, 6, torch.float64), ("rbf", 1, 300, 200, 6, torch.float64)])
def test_multiband_routes_are_deterministic_and_finite_in_every_variant(kind, d, M, N, D, dt):
    """Every kernel variant of the multi-band forward / adjoints (dyadic order x staged dims x ring precision): the same backward six
    times gives the same bits, and they match the oracle.  (A variant with 150 spilled registers returned NaN gradients, one with 330
    registers last-bit run-to-run differences: reads left in flight are now confined to the variants without spills or AGPRs.)"""
    gen = torch.Generator().manual_seed(1)
    A, B = 6, 7
    Xc, Yc = walk(gen, A, M, D, dt), walk(gen, B, N, D, dt)
    w = torch.randn(A, B, generator=gen).to(dt)
    k = sigkernel_amd.RBFKernel(0.9) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, d)
    outs = []
    for _ in range(6):
        Xg = Xc.to(DEV).requires_grad_(True)
        K = sk.compute_Gram(Xg, Yc.to(DEV))
        (K * w.to(DEV)).sum().backward()
        outs.append((K.detach().clone(), Xg.grad.clone()))
    assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
    want = O.gram_grad_weighted(Xc.double(), Yc.double(), w.double().numpy(), k, d, nthreads=NT)
    assert rel_err(outs[0][1].double().cpu().numpy(), want) <= (1e-10 if dt == torch.float64 else 5e-6)


@pytest.mark.gpu
def test_every_route_is_deterministic_run_to_run():
    """tools/det_sweep.py: 600 combinations of static kernel x dyadic order x path dim x lengths x precision x stencil, each through
    compute_Gram + backward, compute_mmd + backward and the derivative Gram, four runs compared bit for bit (no NaN either)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "det_sweep.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "600 combinations x 4 runs, 0 bad" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_every_inline_asm_kernel_family_a_hundred_times():
    """tools/det_sweep.py --families: one shape per hand-scheduled kernel family, 100 runs each, bit for bit -- the stress behind the
    build-time hazard lint (csrc/Makefile): what the lint cannot prove, a race would have to survive 100 times."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "det_sweep.py"), "--families"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "17 combinations x 100 runs, 0 bad" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "rbf"])
def test_paired_batches_of_long_paths_take_the_multiband_adjoints(kind, monkeypatch):
    """compute_kernel(X, Y).backward() on long paths (paired batch, B == 0 in the C ABI) through the multi-band forward with edges and
    the multi-band fused adjoint, fp64 and fp32 inputs, against the oracle pair by pair."""
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(29)
    A, M, N, D, d = 5, 150, 170, 6, 1
    Xc, Yc = walk(gen, A, M, D), walk(gen, A, N, D)
    w = torch.randn(A, generator=gen, dtype=torch.float64)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(0.9)
    for name in ("static_increments", "solve_adj", "static_adjoint"):
        monkeypatch.setattr(type(be), name, (lambda nm: (lambda self, *a, **kw: (_ for _ in ()).throw(AssertionError(nm + " called"))))(name))
    want = np.stack([O.gram_grad_weighted(Xc[i:i + 1], Yc[i:i + 1], w[i:i + 1, None].numpy(), k, d)[0] for i in range(A)])
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 5e-6)):
        Xg = Xc.to(dt).to(DEV).requires_grad_(True)
        kv = sigkernel_amd.SigKernel(k, d).compute_kernel(Xg, Yc.to(dt).to(DEV))
        (kv * w.to(dt).to(DEV)).sum().backward()
        wantd = want if dt == torch.float64 else np.stack([O.gram_grad_weighted(Xc[i:i + 1].float().double(), Yc[i:i + 1].float().double(),
                                                                                  w[i:i + 1, None].float().double().numpy(), k, d)[0] for i in range(A)])
        assert rel_err(Xg.grad.double().cpu().numpy(), wantd) <= tol, (dt, rel_err(Xg.grad.double().cpu().numpy(), wantd))


@pytest.mark.gpu
@pytest.mark.parametrize("screen", [1e3, 1e300])
def test_multiband_fused_adjoint_rescues_an_exploding_pair_on_the_device(screen, monkeypatch):
    """Failure injection on long paths: x_2 and y_5 are the same straight line, k(x_2, y_5) ~ 1e6 and more.  With the screen the
    pair leaves the sweep (residual -1) and its exact stored-grid share is added to its slot on the device; with the screen off it
    fails its self-check and its slot -- node row 0 weights included -- is recomputed.  Every gradient row matches the oracle."""
    be = _lib.get_backend()
    monkeypatch.setattr(type(be), "FUSED_SCREEN", screen)
    gen = torch.Generator().manual_seed(43)
    A, B, M, N, D = 4, 7, 90, 165, 5
    Xc, Yc = walk(gen, A, M, D) * 2, walk(gen, B, N, D) * 2
    Xc[2] = torch.arange(M, dtype=torch.float64)[:, None] * 0.2 * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    Yc[5] = torch.arange(N, dtype=torch.float64)[:, None] * 0.2 * torch.ones(1, D, dtype=torch.float64) / np.sqrt(D)
    k = sigkernel_amd.RBFKernel(2.0)
    w = torch.randn(A, B, generator=gen, dtype=torch.float64)
    Kw = O.gram_forward(Xc, Yc, k, 2, nthreads=NT)
    assert np.sum(np.abs(Kw) > 1e3) == 1 and abs(Kw[2, 5]) > 1e5, np.sort(np.abs(Kw).ravel())[-3:]
    Xg, Yd, wd = Xc.to(DEV).requires_grad_(True), Yc.to(DEV), w.to(DEV)
    torch.cuda.set_sync_debug_mode("warn")
    import warnings
    try:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            (sigkernel_amd.SigKernel(k, 2).compute_Gram(Xg, Yd) * wd).sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert not [r for r in rec if "synchroniz" in str(r.message).lower()]
    err = be.last_fused_err.cpu().numpy().reshape(A, B)
    assert (err[2, 5] == -1.0) if screen < 1e100 else (err[2, 5] > _lib.HipBackend.ADJ_RESIDUAL_TOL or err[2, 5] != err[2, 5])
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, 2, nthreads=NT)
    got = Xg.grad.cpu().numpy()
    for a in range(A):
        assert rel_err(got[a], want[a]) <= 2 * _lib.HipBackend.ADJ_RESIDUAL_TOL, (a, rel_err(got[a], want[a]))


@pytest.mark.gpu
def test_c5_route_with_a_gradient_materialises_nothing(monkeypatch):
    """BASELINE configs[4]'s shape (len 512, dim 16, RBF, dyadic 2, fp32) WITH a gradient: sk_solve_fwd_static_f32 keeps the edges,
    sk_rbf_adjoint_fused_mb_f64 (fp32 ring) sweeps back -- sk_static_increments, sk_solve_adj and sk_static_adjoint are never
    called, nothing of size pairs x M x N exists; the gradient matches the oracle to fp32 resolution (~1e-7) and, on fp64 tensors of
    the same shape, to 5e-10: the backward recompute of K runs from the terminal column across all 2044 fine columns (it restarts
    from exact values at every band of rows, not along a row), and its rounding error grows with that width -- 1e-12 at C4's 252
    columns, 1e-11 at 660, 2e-10 here; north_star asks for 1e-6.  compute_mmd (the symmetric Grams included) goes the same way."""
    be = _lib.get_backend()
    # (fp32 paths -- the config -- take these kernels by default; fp64 tensors of this shape stage 16 fp64 dims, one wave per SIMD,
    # and by default stream where that is faster: the memory-first switch keeps them on the fused route for this test)
    monkeypatch.setattr(sigkernel_amd.routes, "no_stream", True)
    gen = torch.Generator().manual_seed(18)
    X, Y = walk(gen, 3, 512, 16, torch.float32), walk(gen, 4, 512, 16, torch.float32)
    w = torch.randn(3, 4, generator=gen, dtype=torch.float64)
    k = sigkernel_amd.RBFKernel(1.0)
    sk = sigkernel_amd.SigKernel(k, 2)
    boom = lambda name: (lambda self, *a, **kw: (_ for _ in ()).throw(AssertionError(name + " called")))
    for name in ("static_increments", "solve_adj", "static_adjoint", "solve_fwd_keep_edges"):
        monkeypatch.setattr(type(be), name, boom(name))
    want = O.gram_grad_weighted(X.double(), Y.double(), w.numpy(), k, 2, nthreads=NT)
    for dt, tol in ((torch.float32, 2e-6), (torch.float64, 5e-10)):
        Xg = X.to(dt).to(DEV).requires_grad_(True)
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        (sk.compute_Gram(Xg, Y.to(dt).to(DEV)) * w.to(dt).to(DEV)).sum().backward()
        # edges, partial sums, and the rescue's stored grids for 8 pairs at a time (8 x 67 MB, whatever the batch)
        assert torch.cuda.max_memory_allocated() - base < 900 << 20
        assert rel_err(Xg.grad.double().cpu().numpy(), want) <= tol, (dt, rel_err(Xg.grad.double().cpu().numpy(), want))
    Xg = X.double().to(DEV).requires_grad_(True)
    sk.compute_mmd(Xg, Y.double().to(DEV)).backward()
    Xc, Yc = X.double(), Y.double()
    g_xx = 2.0 * O.gram_grad_weighted(Xc, Xc, (np.ones((3, 3)) - np.eye(3)) / 6.0, k, 2, nthreads=NT)   # unbiased: off-diagonal mean, 2x rule
    g_xy = O.gram_grad_weighted(Xc, Yc, np.full((3, 4), -2.0 / 12.0), k, 2, nthreads=NT)
    assert rel_err(Xg.grad.cpu().numpy(), g_xx + g_xy) <= 5e-10


# ---------------------------------------------------------------------------------------------
# the fused RBF adjoint (sk_rbf_adjoint_fused_f64, csrc/sk_wave_adj_fused_rbf.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fused_rbf_adjoint_against_the_unfused_route(monkeypatch):
    """Adjoint PDE + node evaluation + chain rule through the 4-corner difference and the exponential in one kernel, from the
    paths and the forward's terminal edges, against sk_static_increments -> sk_solve_adj -> sk_static_adjoint on 100 random
    shapes (dyadic 1..2, dims 1..8, Gram and paired, with and without an upstream gradient); dL/dX agrees to 1e-10."""
    be = _lib.get_backend()
    rng = np.random.default_rng(1)
    n = 0
    for it in range(100):
        d = int(rng.integers(1, 3))
        cap = 64 * (4 >> d)
        M = int(rng.integers(2, cap + 1)) if it % 4 else cap
        N = int(rng.integers(2, 150))
        A, B, D = int(rng.integers(1, 20)), int(rng.integers(1, 30)), int(rng.integers(1, 9))
        gram = bool(it % 3)
        if not gram:
            B = A
        sig = float(rng.uniform(0.5, 1.5))
        gen = torch.Generator().manual_seed(300 + it)
        X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
        go = torch.randn(A * B if gram else A, generator=gen, dtype=torch.float64).to(DEV) if it % 5 else None
        res = be.solve_fwd_fused_rbf(X, Y, sig, d, False, gram, keep_edges=True)
        assert res is not None and res[1] is not None
        got = be.rbf_adjoint_fused(X, Y, sig, d, res[1], go, gram=gram)
        if got is None:       # node rows / columns that do not fit the edge layout's lanes and units, dims 5..8 at dyadic 2 or beyond 64
            continue          # points at dyadic 1: the multi-band adjoint's / the unfused route's business
        inc = be.static_increments(1, sig, X, Y, gram)
        _, W = be.solve_adj(inc, d, False, edges=res[1])
        want = be.static_adjoint(1, sig, X, Y, W, go, gram)
        assert rel_err(got[0].cpu().numpy(), want.cpu().numpy()) <= max(1e-10, 10 * float(got[1])), (it, d, A, B, M, N, D, gram)
        n += 1
    assert n >= 40


@pytest.mark.gpu
def test_fused_rbf_adjoint_second_argument_sums():
    """The same sweep's SECOND-argument sums (sk_rbf_adjoint_fused_f64 with ypart: per pair and node column of y_b, carried
    down the lanes and stored by the bottom lane) folded with an arbitrary per-pair weight, against sk_static_adjoint2 on the
    unfused route's W, on 60 random shapes (dyadic 1..2, dims 1..4, partial lane groups, b0 > 0); and the first-argument
    gradient of the same launch against the plain one."""
    be = _lib.get_backend()
    rng = np.random.default_rng(7)
    n = 0
    for it in range(60):
        d = int(rng.integers(1, 3))
        cap = 64 * (4 >> d)
        M = int(rng.integers(2, cap + 1)) if it % 4 else cap
        N = int(rng.integers(2, 150)) if it % 5 else M
        A, B, D = int(rng.integers(1, 20)), int(rng.integers(1, 30)), int(rng.integers(1, 5))
        sig = float(rng.uniform(0.5, 1.5))
        gen = torch.Generator().manual_seed(900 + it)
        X, Y = (walk(gen, A, M, D) * 2).to(DEV), (walk(gen, B, N, D) * 2).to(DEV)
        go = torch.randn(A * B, generator=gen, dtype=torch.float64).to(DEV) if it % 3 else None
        w2 = torch.randn(A, B, generator=gen, dtype=torch.float64).to(DEV)
        b0 = int(rng.integers(0, B))
        res = be.solve_fwd_fused_rbf(X, Y, sig, d, False, True, keep_edges=True)
        assert res is not None and res[1] is not None
        got = be.rbf_adjoint_fused(X, Y, sig, d, res[1], go, gram=True, yside=True)
        if got is None:
            continue
        plain = be.rbf_adjoint_fused(X, Y, sig, d, res[1], go, gram=True)
        assert rel_err(got[0].cpu().numpy(), plain[0].cpu().numpy()) <= 1e-12, (it, d, A, B, M, N, D)
        inc = be.static_increments(1, sig, X, Y, True)
        _, W = be.solve_adj(inc, d, False, edges=res[1])
        want = be.static_adjoint2(1, sig, X, Y, W, w2, b0)
        gy = be.second_argument_gradient(got[2], Y, sig, w2, b0)
        assert gy.shape == want.shape
        assert rel_err(gy.cpu().numpy(), want.cpu().numpy()) <= max(1e-10, 10 * float(got[1])), (it, d, A, B, M, N, D, b0)
        n += 1
    assert n >= 30


@pytest.mark.gpu
def test_fused_rbf_adjoint_is_what_the_api_runs(monkeypatch):
    """compute_Gram / compute_kernel gradients with RBFKernel on paths of dim <= 4 go through the fused adjoint and agree with the
    unfused route and with the oracle's closed form; compute_mmd (triangular K_XX + fused K_XY) agrees with the reference fixture."""
    be = _lib.get_backend()
    gen = torch.Generator().manual_seed(31)
    Xc, Yc = walk(gen, 12, 40, 4), walk(gen, 9, 33, 4)      # (N - 1 = 32: the strip is kept one node column wider)
    X, Y = Xc.to(DEV), Yc.to(DEV)
    w = torch.randn(12, 9, generator=gen, dtype=torch.float64)
    k = sigkernel_amd.RBFKernel(0.8)
    sk = sigkernel_amd.SigKernel(k, 2)
    calls = []
    orig = type(be).rbf_adjoint_fused
    monkeypatch.setattr(type(be), "rbf_adjoint_fused", lambda self, *a, **kw: (calls.append(1), orig(self, *a, **kw))[1])
    X1 = X.clone().requires_grad_(True)
    (sk.compute_Gram(X1, Y) * w.to(DEV)).sum().backward()
    assert calls, "RBFKernel backward did not use the fused adjoint"
    want = O.gram_grad_weighted(Xc, Yc, w.numpy(), k, 2, nthreads=8)
    assert rel_err(X1.grad.cpu().numpy(), want) <= 1e-10
    Xp = X[:9].clone().requires_grad_(True)
    sk.compute_kernel(Xp, Y).sum().backward()
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", True)
    X2 = X.clone().requires_grad_(True)
    (sk.compute_Gram(X2, Y) * w.to(DEV)).sum().backward()
    Xq = X[:9].clone().requires_grad_(True)
    sk.compute_kernel(Xq, Y).sum().backward()
    assert rel_err(X1.grad.cpu().numpy(), X2.grad.cpu().numpy()) <= 1e-10
    assert rel_err(Xp.grad.cpu().numpy(), Xq.grad.cpu().numpy()) <= 1e-10
    monkeypatch.setattr(sigkernel_amd.routes, "no_fused_adjoint", False)
    c = golden("gram_c4mini_rbf_d2")
    Xf, Yf = torch.from_numpy(c["X"]).to(DEV), torch.from_numpy(c["Y"]).to(DEV)
    n0 = len(calls)
    Xg = Xf.clone().requires_grad_(True)
    sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"])).compute_mmd(Xg, Yf).backward()
    assert len(calls) > n0
    assert rel_err(Xg.grad.cpu().numpy(), c["grad_mmd"]) <= grad_tol("gram_c4mini_rbf_d2", "grad_mmd")
    # fp32 tensors: swept in fp64 on the up-cast paths
    X3 = X.float().clone().requires_grad_(True)
    (sk.compute_Gram(X3, Y.float()) * w.float().to(DEV)).sum().backward()
    assert X3.grad.dtype == torch.float32 and rel_err(X3.grad.cpu().numpy(), want) <= 2e-4


@pytest.mark.gpu
def test_symmetric_gram_in_one_triangular_launch(monkeypatch):
    """compute_Gram(X, X, sym=True) without a gradient, LinearKernel / RBFKernel within the fused kernels' scope: the pairs on and
    above the diagonal in ONE launch (sk_solve_fwd_*_sym_*), every value written to both halves -- against the oracle, exactly
    symmetric, for batch sizes around the lane-group and wave boundaries; BASELINE configs[1] (batch 128, len 64, dim 3) included."""
    be = _lib.get_backend()
    calls = []
    orig = type(be).solve_fwd_fused_sym
    monkeypatch.setattr(type(be), "solve_fwd_fused_sym", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    gen = torch.Generator().manual_seed(12)
    for A, M, D, d, kern, dt in ((1, 9, 2, 1, sigkernel_amd.RBFKernel(0.7), torch.float64), (2, 30, 3, 0, sigkernel_amd.LinearKernel(), torch.float64),
                                 (7, 64, 8, 2, sigkernel_amd.RBFKernel(1.0), torch.float64), (130, 20, 4, 1, sigkernel_amd.LinearKernel(), torch.float64),
                                 (128, 64, 3, 1, sigkernel_amd.RBFKernel(1.0), torch.float64), (33, 40, 5, 1, sigkernel_amd.RBFKernel(1.3), torch.float32)):
        Xc = walk(gen, A, M, D, dt) * 1.5
        X = Xc.to(DEV)
        n0 = len(calls)
        K = sigkernel_amd.SigKernel(kern, d).compute_Gram(X, X, sym=True)
        assert len(calls) == n0 + 1 and K.shape == (A, A) and K.dtype == dt and torch.equal(K, K.t())
        want = O.gram_forward(Xc.double(), Xc.double(), kern, d, nthreads=NT)
        assert rel_err(K.double().cpu().numpy(), want) <= (1e-11 if dt == torch.float64 else 2e-6), (A, M, D, d)
    # outside the single-band scope the tiled route still answers (two bands at dyadic 1)
    Xl = walk(gen, 5, 300, 3).to(DEV)
    Kl = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(1.0), 1).compute_Gram(Xl, Xl, sym=True)
    assert torch.equal(Kl, Kl.t()) and rel_err(Kl.cpu().numpy(), O.gram_forward(Xl.cpu(), Xl.cpu(), sigkernel_amd.RBFKernel(1.0), 1, nthreads=NT)) <= 1e-11


@pytest.mark.gpu
def test_shares_by_wave_age_rank_do_not_change_any_result(monkeypatch):
    """the launches hand the oldest wave of a SIMD the largest share of the pairs.  Whatever the shares, the pairs
    are partitioned, so forward values (single- and multi-band) are bit-identical to those of equal shares; the fused adjoints
    add their per-chunk partial sums in a different grouping, so gradients agree to rounding (sizes large enough for the shares
    to be in force: >= 12 pairs per resident wave slot; 256 x 256 paths of 64 nodes fill the chip with 8 chunks per row)."""
    gen = torch.Generator().manual_seed(7)
    cases = [
        (sigkernel_amd.LinearKernel(), 1, walk(gen, 384, 64, 8), walk(gen, 400, 64, 8)),        # fused forward, three ranks
        (sigkernel_amd.RBFKernel(0.9), 2, walk(gen, 256, 64, 3), walk(gen, 256, 64, 3)),        # RBF, fused adjoint (chunks by rank)
        (sigkernel_amd.RBFKernel(1.1), 1, walk(gen, 96, 300, 5), walk(gen, 400, 280, 5)),       # multi-band forward
    ]
    for kern, d, Xc, Yc in cases:
        sk = sigkernel_amd.SigKernel(kern, dyadic_order=d)
        X, Y = Xc.to(DEV), Yc.to(DEV)
        w = torch.randn(X.shape[0], Y.shape[0], generator=gen, dtype=torch.float64).to(DEV)
        out = []
        for shares in ("50,50", None, "80,20"):
            for var in ("SK_RANK_W",):
                if shares is None:
                    monkeypatch.delenv(var, raising=False)
                else:
                    monkeypatch.setenv(var, shares if d != 1 or type(kern) is not sigkernel_amd.LinearKernel else {"50,50": "34,33,33", "80,20": "70,20,10"}[shares])
            Xg = X.clone().requires_grad_(True)
            K = sk.compute_Gram(Xg, Y)
            (K * w).sum().backward()
            Kn = sk.compute_Gram(X, Y)
            out.append((K.detach().clone(), Xg.grad.clone(), Kn.clone()))
        for K, g, Kn in out[1:]:
            assert torch.equal(K, out[0][0]) and torch.equal(Kn, out[0][2])
            assert float((g - out[0][1]).abs().max()) <= 1e-12 * float(out[0][1].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,A,B,M,N,D,d,sym", [
    ("rbf", 128, 128, 64, 64, 3, 1, True),     # C2 itself: 8256 pairs = 4096 lane groups x 2 + 64 (two lane groups per wave)
    ("rbf", 100, 103, 14, 23, 2, 1, False),    # eight lane groups per wave, a last wave that is partly empty, 23 columns in 16 padded units
    ("linear", 97, 101, 30, 41, 5, 0, False),  # dim > 4: the eight-dimension slabs
    ("linear", 90, 131, 33, 20, 4, 2, False),  # four lane groups, d = 2
])
def test_small_launches_with_uneven_shares_against_the_oracle(kind, A, B, M, N, D, d, sym):
    """Launches of a few pairs per lane group take no work queue: every lane group gets floor(P / groups) pairs, the first n_big
    waves one more, and a wave ends with its last output store (sk_wave_fused.hip, launch_fused_nd).  Every entry against the
    oracle at 1e-12 (fp64); the entries of the waves with the longer and with the shorter share are both in there."""
    gen = torch.Generator().manual_seed(1000 + A + M)
    Xc, Yc = walk(gen, A, M, D), walk(gen, B, N, D)
    k = sigkernel_amd.RBFKernel(0.8) if kind == "rbf" else sigkernel_amd.LinearKernel()
    sk = sigkernel_amd.SigKernel(k, dyadic_order=d)
    calls = []
    be = _lib.get_backend()
    for name in ("solve_fwd_fused_rbf", "solve_fwd_fused_linear", "solve_fwd_fused_sym"):
        orig = getattr(be, name)
        setattr(be, name, (lambda o, n: (lambda *a, **kw: (calls.append(n), o(*a, **kw))[1]))(orig, name))
    try:
        if sym:
            X = Xc.to(DEV)
            K = sk.compute_Gram(X, X, sym=True)
            want = O.gram_forward(Xc, Xc, k, d, nthreads=NT)
        else:
            K = sk.compute_Gram(Xc.to(DEV), Yc.to(DEV))
            want = O.gram_forward(Xc, Yc, k, d, nthreads=NT)
    finally:
        for name in ("solve_fwd_fused_rbf", "solve_fwd_fused_linear", "solve_fwd_fused_sym"):
            delattr(be, name)
    assert calls, "the single-band fused kernel did not run"
    got = K.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "rbf"])
@pytest.mark.parametrize("screen", [1e3, 1e300])
def test_fused_rescue_finds_the_flagged_pair_among_short_chunks(kind, screen, monkeypatch):
    """140 x 64 pairs of 32 points: more pairs than the fused adjoint has lane groups, so every lane group sweeps a chunk of a few
    pairs -- the case in which k_fused_rescue scans 64 chunks at a time, one per lane (sk_adj_fused_rescue.hip).  One pair explodes;
    marked by the screen or failing its self-check after the fact, its share (or its whole chunk) is re-solved exactly on the device."""
    be = _lib.get_backend()
    monkeypatch.setattr(type(be), "FUSED_SCREEN", screen)
    gen = torch.Generator().manual_seed(43)
    A, B, d = 140, 64, 1
    Xc, Yc = _one_wild_pair(gen, A, B, 32, 4)
    k = sigkernel_amd.LinearKernel() if kind == "linear" else sigkernel_amd.RBFKernel(1.0)
    wc = torch.randn(A, B, generator=gen, dtype=torch.float64)
    Kc = O.gram_forward(Xc, Yc, k, d, nthreads=NT)
    wild = np.abs(Kc) > 1e3
    assert np.abs(Kc[2, 5]) > 1e5 and 1 <= wild.sum() <= 40
    be.last_fused_err = None
    Xg = Xc.to(DEV).requires_grad_(True)
    (sigkernel_amd.SigKernel(k, d).compute_Gram(Xg, Yc.to(DEV)) * wc.to(DEV)).sum().backward()
    assert be.last_fused_err is not None, "the fused adjoint declined the case"
    assert 1 < be.last_fused_ppg < 32, be.last_fused_ppg      # chunks of a few pairs: the lane-per-chunk scan
    err = be.last_fused_err.cpu().numpy().reshape(A, B)
    if screen < 1e100:
        assert np.array_equal(err < 0, wild)
    else:
        assert err[2, 5] > be.ADJ_RESIDUAL_TOL or np.isnan(err[2, 5])
    want = O.gram_grad_weighted(Xc, Yc, wc.numpy(), k, d, nthreads=NT)
    got = Xg.grad.cpu().numpy()
    for a in range(A):
        assert rel_err(got[a], want[a]) <= 2 * be.ADJ_RESIDUAL_TOL, (a, rel_err(got[a], want[a]))
