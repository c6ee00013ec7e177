"""Row-sharded Gram matrices over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every (x_i, y_j) pair is an independent PDE, so the Gram matrix shards by ROWS of X with no exchange
during the solve; the reference has no multi-GPU code at all.  Rank r of R owns
rows [r*ceil(A/R), (r+1)*ceil(A/R)) of X, Y is replicated (it is tiny), and ONE all-gather of the
(rows x B) value blocks gives every rank the full matrix -- a few MB per rank, latency-bound on the
point-to-point xGMI links, so no bucketing or ring tuning is warranted.  In backward the gradient rows are
rank-local too (row a of grad_X needs only row a of grad_output, also under the reference's 2x rule), and
a second all-gather returns the full grad_X.  A symmetric Gram without a gradient (compute_mmd's K_YY) is solved on the
triangle, folded over the ranks so that the load stays even (`_folded_symmetric_gram`).

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist

from . import _lib
from .sigkernel import (_SigKernel, _SigKernelGram, _budget, _fused_static, _gram_block, _sym_fused_gradient, _sym_triangle_ok,
                        _sym_unfused_gradient, k_kgrad)

__all__ = ["row_range", "sharded_gram", "sharded_kernel", "ShardedGram", "ShardedPaired", "ShardedSymGram", "sharded_kgrad",
           "record_collectives", "collective_summary"]

# Per-collective timings, on request (bench.py's N > 1 line: what a step spends in RCCL as opposed to its kernels).  While a list
# hangs here every collective of this module is bracketed by two events on the caller's stream (HIP events on a GPU -- a
# torch.distributed collective makes the caller's stream wait for it, so the second event completes when the collective has --,
# the wall clock on the CPU / gloo test configuration).  Off (None) by default: nothing is recorded, nothing synchronises.
_COLLECTIVE_LOG = None


def record_collectives(on=True):
    """Start (or stop and drop) the log of this module's collectives; collective_summary() reads it."""
    global _COLLECTIVE_LOG
    _COLLECTIVE_LOG = [] if on else None


def collective_summary():
    """{"all_gather": {"calls", "ms", "bytes"}, "all_reduce": {...}} of the collectives logged since record_collectives(); synchronises
    the device.  bytes: what this rank RECEIVES (all-gather: the gathered tensor; all-reduce: the reduced tensor)."""
    log = _COLLECTIVE_LOG or []
    if any(e[2] is not None and not isinstance(e[2], float) for e in log):
        torch.cuda.synchronize()
    out = {}
    for kind, nbytes, t0, t1 in log:
        ms = (t1 - t0) * 1e3 if isinstance(t0, float) else t0.elapsed_time(t1)
        ent = out.setdefault(kind, {"calls": 0, "ms": 0.0, "bytes": 0})
        ent["calls"] += 1
        ent["ms"] += float(ms)
        ent["bytes"] += int(nbytes)
    return out


class _timed_collective:
    def __init__(self, kind, tensor):
        self.on = _COLLECTIVE_LOG is not None
        self.kind, self.tensor = kind, tensor

    def __enter__(self):
        if self.on:
            import time as _t
            if self.tensor.is_cuda:
                self.t0 = torch.cuda.Event(enable_timing=True)
                self.t0.record()
            else:
                self.t0 = _t.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.on and exc[0] is None and _COLLECTIVE_LOG is not None:
            import time as _t
            if self.tensor.is_cuda:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
            else:
                t1 = _t.perf_counter()
            _COLLECTIVE_LOG.append((self.kind, self.tensor.numel() * self.tensor.element_size(), self.t0, t1))
        return False


def row_range(n_rows, rank, world):
    """Rows owned by `rank`: equal chunks of ceil(n/world), the last ranks may own fewer (or none)."""
    chunk = -(-n_rows // world)
    lo = min(rank * chunk, n_rows)
    return lo, min(lo + chunk, n_rows), chunk


def _gather(out, inp, group):
    """all_gather_into_tensor; a gloo group is given host copies of device tensors (gloo moves no HIP memory: this is the
    test configuration -- several ranks sharing one GPU -- not a production path, which is RCCL)."""
    with _timed_collective("all_gather", out):
        if out.is_cuda and dist.get_backend(group) == "gloo":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, inp.cpu(), group=group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp, group=group)


def _all_gather_rows(block, n_rows, chunk, group):
    """block: this rank's rows, shape (rows_r, ...) with rows_r <= chunk -> (n_rows, ...) on every rank."""
    world = dist.get_world_size(group)
    if block.shape[0] == chunk:          # the rows divide evenly (the usual case): the block goes out as it is -- at 64 rows per rank
        pad = block.contiguous()         # the zero-fill and the copy were two launches of a 0.6 ms step
    else:
        pad = torch.zeros((chunk,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        pad[: block.shape[0]] = block
    out = torch.empty((world * chunk,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    _gather(out, pad, group)
    return out[:n_rows]


def _folded_symmetric_gram(X, static_kernel, dyadic_order, naive, workspace_bytes, group):
    """compute_Gram(X, X, sym=True) without a gradient, sharded: only the pairs on and above the diagonal are solved, like the
    reference's CPU solver does (cython_backend.pyx:74-97).  The rows are cut into 2R blocks and rank r takes blocks r and
    2R-1-r, each against the columns from its own first row on: every rank solves (2R+1)/(2R)^2 of the square, half of its
    row-shard, and the load is even.  One all-gather of the two zero-padded strips per rank, then the mirror."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    A = X.shape[0]
    bs = -(-A // (2 * world))
    strips = torch.zeros(2, bs, A, dtype=X.dtype, device=X.device)
    Xd = X.detach()
    with torch.no_grad():
        for slot, blk in enumerate((rank, 2 * world - 1 - rank)):
            lo, hi = min(blk * bs, A), min((blk + 1) * bs, A)
            if hi > lo:
                strips[slot, : hi - lo, lo:] = _SigKernelGram.apply(Xd[lo:hi].contiguous(), Xd[lo:].contiguous(), static_kernel,
                                                                    dyadic_order, False, naive, workspace_bytes)
    return _assemble_folded(strips, A, bs, world, group)


class ShardedGram(torch.autograd.Function):
    """compute_Gram with the rows of X sharded over a process group; returns the full matrix on every rank."""

    @staticmethod
    def forward(ctx, X, Y, static_kernel, dyadic_order, sym, _naive_solver, workspace_bytes, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        A = X.shape[0]
        lo, hi, chunk = row_range(A, rank, world)
        ctx.args = (group, A)
        ctx.local = None
        ctx.meta = (tuple(X.shape[1:]), X.dtype, X.device)
        if hi > lo:
            # the local block goes through the single-GPU Function with its graph kept: when a gradient is pending its forward
            # keeps the terminal edges of every pair, so that backward runs the adjoint sweep only (no second forward solve)
            Xl = X.detach()[lo:hi].contiguous().requires_grad_(X.requires_grad)
            with torch.enable_grad():
                Kl = _SigKernelGram.apply(Xl, Y.detach(), static_kernel, dyadic_order, False, _naive_solver, workspace_bytes)
            if X.requires_grad:
                ctx.local = (Xl, Kl)
            Kloc = Kl.detach()
        else:
            Kloc = torch.empty((0, Y.shape[0]), dtype=X.dtype, device=X.device)
        return _all_gather_rows(Kloc, A, chunk, group)

    @staticmethod
    def backward(ctx, grad_output):
        group, A = ctx.args
        tail, dtype, device = ctx.meta
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi, chunk = row_range(A, rank, world)
        if hi > lo and ctx.local is not None:
            # the local graph is kept (retain_graph): a second backward through the same graph (retain_graph=True upstream,
            # repeated autograd.grad) must give the same rows again -- the inner Function drops the kept edges after their
            # first use and re-sweeps forward by itself from then on
            Xl, Kl = ctx.local
            (gl,) = torch.autograd.grad(Kl, Xl, grad_output[lo:hi].to(dtype), retain_graph=True)
        else:
            if hi > lo and ctx.needs_input_grad[0]:
                raise RuntimeError("ShardedGram.backward: the local graph of this rank's rows is gone")
            gl = torch.zeros((hi - lo,) + tail, dtype=dtype, device=device)
        grad_X = _all_gather_rows(gl, A, chunk, group)
        if ctx.needs_input_grad[1]:      # the reference's 2x rule (sigkernel.py:410-412)
            grad_X = 2 * grad_X
        return grad_X, None, None, None, None, None, None, None


class ShardedPaired(torch.autograd.Function):
    """compute_kernel (the paired batch k(x_i, y_i), sigkernel.py:23-40 / _SigKernel :201-343) with the P = A pairs sharded over a
    process group exactly like Gram rows: rank r solves pairs [r ceil(A/R), ...) with no data-path collective, ONE all-gather of the
    (A/R,) values gives every rank the full vector; in backward the gradient rows are rank-local (row i needs only grad_output[i])
    and a second all-gather of the (A/R, M, D) rows returns the full grad_X.  No gradient for Y, as in the reference (:343)."""

    @staticmethod
    def forward(ctx, X, Y, static_kernel, dyadic_order, _naive_solver, workspace_bytes, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        A = X.shape[0]
        lo, hi, chunk = row_range(A, rank, world)
        ctx.args = (group, A)
        ctx.local = None
        ctx.meta = (tuple(X.shape[1:]), X.dtype, X.device)
        if hi > lo:
            Xl = X.detach()[lo:hi].contiguous().requires_grad_(X.requires_grad)
            with torch.enable_grad():      # the local graph is kept: its forward keeps the edges, backward is one adjoint launch
                Kl = _SigKernel.apply(Xl, Y.detach()[lo:hi].contiguous(), static_kernel, dyadic_order, _naive_solver, workspace_bytes)
            if X.requires_grad:
                ctx.local = (Xl, Kl)
            Kloc = Kl.detach()
        else:
            Kloc = torch.empty((0,), dtype=X.dtype, device=X.device)
        return _all_gather_rows(Kloc, A, chunk, group)

    @staticmethod
    def backward(ctx, grad_output):
        group, A = ctx.args
        tail, dtype, device = ctx.meta
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi, chunk = row_range(A, rank, world)
        if hi > lo and ctx.local is not None:
            Xl, Kl = ctx.local
            (gl,) = torch.autograd.grad(Kl, Xl, grad_output[lo:hi].to(dtype), retain_graph=True)
        else:
            if hi > lo and ctx.needs_input_grad[0]:
                raise RuntimeError("ShardedPaired.backward: the local graph of this rank's pairs is gone")
            gl = torch.zeros((hi - lo,) + tail, dtype=dtype, device=device)
        return _all_gather_rows(gl, A, chunk, group), None, None, None, None, None, None


def sharded_kernel(sigkernel, X, Y, group=None):
    """Full (A,) vector k(x_i, y_i) on every rank; each rank solves only its pairs."""
    if not dist.is_initialized():
        raise RuntimeError("sharded_kernel needs torch.distributed to be initialised (one process per GPU)")
    from .sigkernel import _check_inputs
    _check_inputs(X, Y, paired=True)
    return ShardedPaired.apply(X, Y, sigkernel.static_kernel, sigkernel.dyadic_order, sigkernel._naive_solver, sigkernel.workspace_bytes, group)


def _folded_blocks(A, rank, world):
    """The two row blocks of rank `rank` when the rows are cut into 2 x world blocks and folded: blocks r and 2R-1-r."""
    bs = -(-A // (2 * world))
    return bs, [(min(blk * bs, A), min((blk + 1) * bs, A)) for blk in (rank, 2 * world - 1 - rank)]


def _assemble_folded(strips, A, bs, world, group):
    """All-gather of every rank's two zero-padded strips (2, bs, A) -> the full, exactly symmetric (A, A) matrix."""
    full = torch.empty(world * 2, bs, A, dtype=strips.dtype, device=strips.device)     # rank-major concatenation along dim 0
    _gather(full, strips, group)
    full = full.reshape(world, 2, bs, A)
    K = torch.empty(2 * world * bs, A, dtype=strips.dtype, device=strips.device)
    for r in range(world):
        K[r * bs:(r + 1) * bs] = full[r, 0]
        K[(2 * world - 1 - r) * bs:(2 * world - r) * bs] = full[r, 1]
    K = K[:A]
    iu = torch.triu_indices(A, A, offset=1, device=strips.device)
    K[iu[1], iu[0]] = K[iu[0], iu[1]]
    return K


def _all_reduce_sum(t, group):
    with _timed_collective("all_reduce", t):
        if t.is_cuda and dist.get_backend(group) == "gloo":      # (test configuration: several ranks sharing one GPU)
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class ShardedSymGram(torch.autograd.Function):
    """compute_Gram(X, X, sym=True) WITH a gradient (compute_mmd's K_XX, sigkernel.py:190) over a process group: the triangle,
    folded over the ranks like `_folded_symmetric_gram` -- rank r solves row blocks r and 2R-1-r against the columns from the
    block's first row on, (2R+1)/(2R)^2 of the square.  In backward a solved pair (a, b) gives its first-argument gradient to
    row a and, through the second-argument sums of the same adjoint sweep, what the unsolved mirror pair (b, a) owes to row b
    (sigkernel._sym_fused_gradient); rows of grad_X therefore receive contributions from several ranks, and the path has a
    real exchange step: ONE all-reduce (sum) of the (A, M, D) gradient -- 4 MB at BASELINE configs[3] -- over RCCL."""

    @staticmethod
    def forward(ctx, X, static_kernel, dyadic_order, _naive_solver, workspace_bytes, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        be = _lib.get_backend()
        A = X.shape[0]
        bs, blocks = _folded_blocks(A, rank, world)
        Xd = X.detach().contiguous()
        strips = torch.zeros(2, bs, A, dtype=X.dtype, device=X.device)
        kept_blocks = []
        for slot, (lo, hi) in enumerate(blocks):
            if hi > lo:
                kept = []
                strips[slot, : hi - lo, lo:] = _gram_block(be, static_kernel, Xd[lo:hi].contiguous(), Xd[lo:].contiguous(),
                                                           dyadic_order, _naive_solver, workspace_bytes, 3, kept)
                kept_blocks.append((lo, hi, kept))
        ctx.save_for_backward(X)
        ctx.args = (static_kernel, dyadic_order, _naive_solver, workspace_bytes, group)
        ctx.blocks = kept_blocks
        K = _assemble_folded(strips, A, bs, world, group)
        ctx.K = K.detach()        # forward values: what arms the fused adjoint's device-side rescue
        return K

    @staticmethod
    def backward(ctx, grad_output):
        (X,) = ctx.saved_tensors
        static_kernel, d, naive, workspace_bytes, group = ctx.args
        be = _lib.get_backend()
        Xd = X.detach().contiguous()
        go = grad_output.to(X.dtype).contiguous()
        budget = _budget(X.device, workspace_bytes)
        grad = _sym_fused_gradient(be, static_kernel, Xd, go, d, naive, ctx.blocks, budget, getattr(ctx, "K", None))
        if grad is None:
            kind, param = _fused_static(static_kernel, True)
            grad = _sym_unfused_gradient(be, kind, param, Xd, go, d, naive, ctx.blocks, budget)
        ctx.blocks = [(lo, hi, []) for lo, hi, _ in ctx.blocks]      # a second backward sweeps forward again by itself
        grad = _all_reduce_sum(grad.contiguous(), group)
        # X is both arguments: the reference's 2x rule (sigkernel.py:410-412)
        return 2 * grad, None, None, None, None, None


def sharded_gram(sigkernel, X, Y, sym=False, group=None):
    """Full (A, B) Gram matrix on every rank; each rank solves only its rows."""
    if not dist.is_initialized():
        raise RuntimeError("sharded_gram needs torch.distributed to be initialised (one process per GPU)")
    same = X.shape == Y.shape and X.data_ptr() == Y.data_ptr() and X.stride() == Y.stride()
    if dist.get_world_size(group) == 1 and sym and same:      # nothing to fold: the single-GPU Function owns the symmetric shortcuts
        return _SigKernelGram.apply(X, Y, sigkernel.static_kernel, sigkernel.dyadic_order, sym, sigkernel._naive_solver,
                                    sigkernel.workspace_bytes)
    if (sym and same and not X.requires_grad and dist.get_world_size(group) > 1 and X.shape[0] >= 2 and X.shape[1] >= 2):
        return _folded_symmetric_gram(X, sigkernel.static_kernel, sigkernel.dyadic_order, sigkernel._naive_solver,
                                      sigkernel.workspace_bytes, group)
    be = _lib.get_backend()
    sk = sigkernel.static_kernel
    if (sym and same and X.requires_grad and dist.get_world_size(group) > 1 and X.shape[0] >= 2 and X.shape[1] >= 2
            and _sym_triangle_ok(be, sk, X.detach(), sigkernel.dyadic_order, sigkernel._naive_solver)):
        # the triangle WITH a gradient: folded row blocks, second-argument sums, one all-reduce of the gradient
        return ShardedSymGram.apply(X, sk, sigkernel.dyadic_order, sigkernel._naive_solver, sigkernel.workspace_bytes, group)
    return ShardedGram.apply(X, Y, sk, sigkernel.dyadic_order, sym, sigkernel._naive_solver,
                             sigkernel.workspace_bytes, group)


def sharded_kgrad(sigkernel, X, Y, gamma, group=None):
    """compute_kernel_and_derivatives_Gram with the rows of (X, gamma) sharded: three full (A, B) matrices on every
    rank from one all-gather of the stacked (rows, 3, B) blocks."""
    if not dist.is_initialized():
        raise RuntimeError("sharded_kgrad needs torch.distributed to be initialised (one process per GPU)")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    A, B = X.shape[0], Y.shape[0]
    lo, hi, chunk = row_range(A, rank, world)
    if hi > lo:
        loc = torch.stack(k_kgrad(X[lo:hi].contiguous(), Y, gamma[lo:hi].contiguous(), sigkernel.dyadic_order,
                                  sigkernel.static_kernel, workspace_bytes=sigkernel.workspace_bytes), dim=1)
    else:
        loc = torch.empty((0, 3, B), dtype=X.dtype, device=X.device)
    full = _all_gather_rows(loc, A, chunk, group)
    return full[:, 0].contiguous(), full[:, 1].contiguous(), full[:, 2].contiguous()
