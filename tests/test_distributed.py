"""Row-sharded Gram over 2 ranks (gloo, CPU) with the oracle-backed fake solver: the N>1 path of
sigkernel_amd.distributed must reproduce the single-process results and the reference fixtures."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden, grad_tol, make_kernel, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sigkernel_amd
        from sigkernel_amd import _lib
        from fake_backend import OracleBackend
        _lib.set_backend(OracleBackend())
        c = golden(name)
        X, Y, w = (torch.from_numpy(c[k]) for k in ("X", "Y", "w"))
        sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]),
                                     process_group=dist.group.WORLD)
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        (K * w).sum().backward(retain_graph=True)
        res = {"gram": K.detach().numpy(), "grad_w": Xg.grad.numpy().copy()}
        # a second backward through the same graph must give the same gradient again (not zeros)
        (g2,) = torch.autograd.grad((K * w).sum(), Xg)
        res["grad_w_again"] = g2.numpy()
        if "mmd" in c:
            Xg = X.clone().requires_grad_(True)
            mmd = sk.compute_mmd(Xg, Y)
            mmd.backward()
            res["mmd"] = mmd.detach().numpy()
            res["grad_mmd"] = Xg.grad.numpy()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("gram_c2mini_rbf_d1", 2), ("gram_c3mini_lin_d1", 2), ("gram_lin_d0_ragged", 2),
                                        ("gram_len2", 2), ("gram_lin_d2_ragged", 4)])
def test_sharded_gram_matches_reference(tmp_path, name, world):
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    c = golden(name)
    for r in range(world):
        got = dict(np.load(tmp_path / ("rank%d.npz" % r)))
        assert rel_err(got["gram"], c["gram"]) <= 1e-13          # every rank holds the full matrix
        assert rel_err(got["grad_w"], c["grad_w"]) <= grad_tol(name, "grad_w")    # the reference's own FD noise, per fixture
        assert np.array_equal(got["grad_w_again"], got["grad_w"])
        if "mmd" in c:
            assert abs(float(got["mmd"]) - float(c["mmd"])) <= 1e-12
            assert rel_err(got["grad_mmd"], c["grad_mmd"]) <= grad_tol(name, "grad_mmd")


def test_row_range_partitions_all_rows():
    from sigkernel_amd.distributed import row_range
    for n in (1, 5, 8, 13, 512):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                lo, hi, chunk = row_range(n, r, world)
                assert hi - lo <= chunk
                rows += list(range(lo, hi))
            assert rows == list(range(n))


def _paired_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sigkernel_amd
        from sigkernel_amd import _lib
        from fake_backend import OracleBackend
        _lib.set_backend(OracleBackend())
        c = golden(name)
        n = c["paired"].shape[0]
        X, Y, wp = torch.from_numpy(c["X"][:n]), torch.from_numpy(c["Y"][:n]), torch.from_numpy(c["wp"])
        sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]), process_group=dist.group.WORLD)
        Xg = X.clone().requires_grad_(True)
        Kp = sk.compute_kernel(Xg, Y)
        (g1,) = torch.autograd.grad((Kp * wp).sum(), Xg, retain_graph=True)
        (g2,) = torch.autograd.grad((Kp * wp).sum(), Xg)
        res = {"paired": Kp.detach().numpy(), "grad_paired": g1.numpy(), "grad_again": g2.numpy()}
        if X.shape == Y.shape:
            Xd = X.clone().requires_grad_(True)
            dd = sk.compute_distance(Xd, Y)
            dd.backward()
            one = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]))
            Xe = X.clone().requires_grad_(True)
            de = one.compute_distance(Xe, Y)
            de.backward()
            res.update(dist=dd.detach().numpy(), dist_one=de.detach().numpy(), gdist=Xd.grad.numpy(), gdist_one=Xe.grad.numpy())
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("gram_c2mini_rbf_d1", 2), ("gram_c2mini_rbf_d1", 3), ("gram_lin_d0_ragged", 2), ("gram_lin_d0_ragged", 4)])
def test_sharded_paired_batch_matches_reference(tmp_path, name, world):
    """compute_kernel / compute_distance under a process group: the P = A pairs shard over the ranks like Gram rows (one all-gather of
    the values, one of the gradient rows; sigkernel.py:23-40) -- every rank ends with the reference's full vector and gradient, also
    when some ranks own no pair (3 pairs on 4 ranks)."""
    mp.spawn(_paired_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    c = golden(name)
    for r in range(world):
        got = dict(np.load(tmp_path / ("rank%d.npz" % r)))
        assert rel_err(got["paired"], c["paired"]) <= 1e-13
        assert rel_err(got["grad_paired"], c["grad_paired"]) <= grad_tol(name, "grad_paired")
        assert np.array_equal(got["grad_again"], got["grad_paired"])
        if "dist" in got:
            assert abs(float(got["dist"]) - float(got["dist_one"])) <= 1e-13 and rel_err(got["gdist"], got["gdist_one"]) <= 1e-12


def _nccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import sigkernel_amd
        c = golden("gram_c3mini_lin_d1")
        X, Y, w = (torch.from_numpy(c[k]).cuda() for k in ("X", "Y", "w"))
        sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), process_group=dist.group.WORLD)
        Xg = X.clone().requires_grad_(True)
        K = sk.compute_Gram(Xg, Y)
        (K * w).sum().backward()
        g = torch.randn(X.shape, dtype=X.dtype, generator=torch.Generator().manual_seed(1)).cuda()
        kk = sk.compute_kernel_and_derivatives_Gram(X, Y, g)
        k1 = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"])).compute_kernel_and_derivatives_Gram(X, Y, g)
        np.savez(os.path.join(out_dir, "nccl.npz"), gram=K.detach().cpu().numpy(), grad_w=Xg.grad.cpu().numpy(),
                 kd_err=float(max((a - b).abs().max() for a, b in zip(kk, k1))))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_gram_over_rccl_single_rank(tmp_path):
    """The RCCL path of sigkernel_amd.distributed on a real GPU (one rank is all a 1-GPU box offers): process-group init,
    all_gather_into_tensor of the value and gradient blocks, and the sharded derivative Gram."""
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    c = golden("gram_c3mini_lin_d1")
    got = dict(np.load(tmp_path / "nccl.npz"))
    assert rel_err(got["gram"], c["gram"]) <= 1e-12
    assert rel_err(got["grad_w"], c["grad_w"]) <= grad_tol("gram_c3mini_lin_d1", "grad_w")
    assert float(got["kd_err"]) == 0.0


def _sym_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sigkernel_amd
        from sigkernel_amd import _lib, distributed
        from fake_backend import OracleBackend
        _lib.set_backend(OracleBackend())
        calls = []
        orig = distributed._SigKernelGram.apply
        distributed._SigKernelGram.apply = staticmethod(lambda X, Y, *a: (calls.append(X.shape[0] * Y.shape[0]), orig(X, Y, *a))[1])
        gen = torch.Generator().manual_seed(3)
        X = torch.cumsum(torch.randn(11, 7, 2, generator=gen, dtype=torch.float64), dim=1) * 0.3
        sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(0.7), 1, process_group=dist.group.WORLD)
        K = sk.compute_Gram(X, X, sym=True)
        np.savez(os.path.join(out_dir, "sym%d.npz" % rank), gram=K.numpy(), pairs=np.array(sum(calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_symmetric_gram_is_folded_over_the_ranks(tmp_path, world):
    """compute_Gram(X, X, sym=True) without a gradient under a process group: every rank solves about half of its row shard (the
    pairs on and above the diagonal, folded so that the load is even) and all ranks end with the full, exactly symmetric matrix."""
    mp.spawn(_sym_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import sigkernel_amd
    from oracle import oracle as O
    gen = torch.Generator().manual_seed(3)
    X = torch.cumsum(torch.randn(11, 7, 2, generator=gen, dtype=torch.float64), dim=1) * 0.3
    want = O.gram_forward(X, X, sigkernel_amd.RBFKernel(0.7), 1)
    solved = []
    for r in range(world):
        got = dict(np.load(tmp_path / ("sym%d.npz" % r)))
        assert rel_err(got["gram"], want) <= 1e-13 and np.array_equal(got["gram"], got["gram"].T)
        solved.append(int(got["pairs"]))
    assert max(solved) <= 0.75 * (11 * 11 / world) + 11 and sum(solved) < 0.75 * 11 * 11, solved


def _wide_paths():
    gen = torch.Generator().manual_seed(12)
    mk = lambda A: torch.cumsum(torch.randn(A, 20, 20, generator=gen, dtype=torch.float64), dim=1) * 0.05
    return mk(7), mk(5)


def _hip_gloo_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sigkernel_amd
        res = {}
        for name in ("gram_c4mini_rbf_d2", "gram_c3mini_lin_d1"):
            c = golden(name)
            X, Y, w = (torch.from_numpy(c[k]).cuda() for k in ("X", "Y", "w"))
            sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), process_group=dist.group.WORLD)
            Xg = X.clone().requires_grad_(True)
            K = sk.compute_Gram(Xg, Y)
            (K * w).sum().backward()
            res[name + ".gram"], res[name + ".grad_w"] = K.detach().cpu().numpy(), Xg.grad.cpu().numpy()
            Xg = X.clone().requires_grad_(True)
            mmd = sk.compute_mmd(Xg, Y)
            mmd.backward()
            res[name + ".mmd"], res[name + ".grad_mmd"] = mmd.detach().cpu().numpy(), Xg.grad.cpu().numpy()
            # the paired batch, sharded like Gram rows (ShardedPaired)
            n = c["paired"].shape[0]
            Xg = X[:n].clone().requires_grad_(True)
            Kp = sk.compute_kernel(Xg, Y[:n])
            (Kp * torch.from_numpy(c["wp"]).cuda()).sum().backward()
            res[name + ".paired"], res[name + ".grad_paired"] = Kp.detach().cpu().numpy(), Xg.grad.cpu().numpy()
        # LinearKernel with path dim 9..32 and a gradient under a process group -- compute_mmd's K_XX is
        # compute_Gram(X, X, sym=True); the triangle has no second-argument kernel for such paths and must not be chosen
        X12, Y12 = _wide_paths()
        for D in (12, 20):
            sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 1, process_group=dist.group.WORLD)
            Xg = X12[:, :, :D].contiguous().cuda().requires_grad_(True)
            mmd = sk.compute_mmd(Xg, Y12[:, :, :D].contiguous().cuda())
            mmd.backward()
            res["lin%d.mmd" % D], res["lin%d.grad_mmd" % D] = mmd.detach().cpu().numpy(), Xg.grad.cpu().numpy()
        np.savez(os.path.join(out_dir, "hipgloo%d.npz" % rank), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_gram_two_ranks_sharing_one_gpu(tmp_path):
    """World size 2 with the HIP kernels: two processes share the one GPU of the box, each solves its row shard on it, the
    collectives go through gloo (host copies).  RCCL cannot be used for this (it refuses two ranks on one device); the RCCL
    path itself is covered by the single-rank test above and bench.py --force-dist."""
    from conftest import grad_tol
    mp.spawn(_hip_gloo_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = dict(np.load(tmp_path / ("hipgloo%d.npz" % r)))
        for name in ("gram_c4mini_rbf_d2", "gram_c3mini_lin_d1"):
            c = golden(name)
            assert rel_err(got[name + ".gram"], c["gram"]) <= 1e-11
            assert rel_err(got[name + ".grad_w"], c["grad_w"]) <= grad_tol(name, "grad_w")
            assert abs(float(got[name + ".mmd"]) - float(c["mmd"])) <= 1e-11
            assert rel_err(got[name + ".grad_mmd"], c["grad_mmd"]) <= grad_tol(name, "grad_mmd")
            assert rel_err(got[name + ".paired"], c["paired"]) <= 1e-11
            assert rel_err(got[name + ".grad_paired"], c["grad_paired"]) <= grad_tol(name, "grad_paired")
        import sigkernel_amd
        from oracle import oracle as O
        X12, Y12 = _wide_paths()
        for D in (12, 20):
            X, Y, k = X12[:, :, :D].contiguous(), Y12[:, :, :D].contiguous(), sigkernel_amd.LinearKernel()
            A, B = X.shape[0], Y.shape[0]
            Kxx, Kyy, Kxy = O.gram_forward(X, X, k, 1), O.gram_forward(Y, Y, k, 1), O.gram_forward(X, Y, k, 1)
            mmd = (Kxx.sum() - np.trace(Kxx)) / (A * (A - 1.0)) + (Kyy.sum() - np.trace(Kyy)) / (B * (B - 1.0)) - 2.0 * Kxy.mean()
            gw = 2.0 * O.gram_grad_weighted(X, X, (1.0 - np.eye(A)) / (A * (A - 1.0)), k, 1) + \
                O.gram_grad_weighted(X, Y, np.full((A, B), -2.0 / (A * B)), k, 1)
            assert abs(float(got["lin%d.mmd" % D]) - mmd) <= 1e-11 and rel_err(got["lin%d.grad_mmd" % D], gw) <= 1e-9, D
