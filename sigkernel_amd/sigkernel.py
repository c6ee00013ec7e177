"""SigKernel API and autograd surface, mirroring the reference's sigkernel/sigkernel.py:15-416.

Host code only: static kernels stay torch ops (rocBLAS under PyTorch-ROCm); increments, the
Goursat PDE solve and the adjoint PDE run in the HIP kernels of libsigkernel_amd.so through
:mod:`sigkernel_amd._lib`.  Differences from the reference that are not observable in results:

* the refined increment tensor of ``tile()`` (sigkernel.py:218, :364) is never built;
* the adjoint is solved lazily in ``backward`` for both Functions (the reference does it eagerly in
  ``_SigKernelGram.forward``, sigkernel.py:397-399) and contracted with the *analytic* derivative
  of the static kernel instead of the h = 1e-9 finite difference (sigkernel.py:313-341, :472-500);
* big batches are tiled by an HBM budget (``SigKernel.workspace_bytes``), not by ``max_batch``:
  results never depended on ``max_batch`` (sigkernel.py:31-39, :102-127) and still do not.
"""
import functools

import torch

from . import _lib
from ._routes import routes
from .static_kernels import LinearKernel, RBFKernel

__all__ = ["SigKernel", "_SigKernel", "_SigKernelGram", "k_kgrad"]

_DEFAULT_WORKSPACE = 48 << 30  # bytes of transient HBM one call may use (288 GB part)


def _budget(device, requested):
    if requested is not None:
        return int(requested)
    if device.type == "cuda":
        free, _ = torch.cuda.mem_get_info(device)
        return int(min(_DEFAULT_WORKSPACE, 0.5 * free))
    return _DEFAULT_WORKSPACE


def _tiles(n_rows, bytes_per_row, budget):
    rows = int(max(1, min(n_rows, budget // max(1, bytes_per_row))))
    return [(a, min(a + rows, n_rows)) for a in range(0, n_rows, rows)]


def _cap_rows(bytes_per_row, pairs_per_row, budget):
    """bytes_per_row, raised where needed so that a row tile of `budget` bytes never holds more than _MAX_LAUNCH_PAIRS pairs."""
    return max(bytes_per_row, -(-int(budget) * int(pairs_per_row) // _MAX_LAUNCH_PAIRS))


def _fused_static(static_kernel, gram):
    """(kind, param) when the static kernel is exactly one of the two the fused HIP kernels implement
    (sk_static_increments_*: static kernel + increments in one pass, G_static never materialised), else None.
    Subclasses and user-defined kernels always take the generic Gram_matrix / batch_kernel path."""
    if type(static_kernel) is LinearKernel:
        # the reference's Gram_matrix ignores `scale`, batch_kernel applies it to both arguments (static_kernels.py:24,33)
        return 0, (1.0 if gram else float(static_kernel.scale))
    if type(static_kernel) is RBFKernel and float(static_kernel.sigma) > 0:
        return 1, float(static_kernel.sigma)
    return None


STREAM, FUSED, FUSED_MB, FUSED_MB_SWAP = _lib.ROUTE_STREAM, _lib.ROUTE_FUSED, _lib.ROUTE_FUSED_MB, _lib.ROUTE_FUSED_MB_SWAP
FUSED_SWAP = _lib.ROUTE_FUSED_SWAP
OP_FORWARD, OP_ADJOINT, OP_ADJOINT_SYM = _lib.OP_FORWARD, _lib.OP_ADJOINT, _lib.OP_ADJOINT_SYM


@functools.lru_cache(maxsize=4096)
def _route_query(route_fn, *key):
    """sk_route_query is a pure function of its arguments: one ctypes call (1.4 us) per distinct shape, a dictionary look-up after."""
    return route_fn(*key)


def _route(be, op, static_kernel, Xd, Yd, dyadic, naive, gram):
    """Which kernel family serves the call: the library's own answer (sk_route_query, csrc/sk_route.hip -- the one statement of the
    fused kernels' scope) for exactly LinearKernel / exactly RBFKernel, STREAM for every other static kernel; then the route
    switches (`sigkernel_amd.routes`: A/B measurements and the tests that compare a fused route with the streaming one)."""
    fused = _fused_static(static_kernel, gram)
    if fused is None or not hasattr(be, "route"):
        return STREAM
    if (fused[0] == 1 and routes.no_fused_rbf) or (op == OP_ADJOINT and routes.no_fused_adjoint):
        return STREAM
    # (a swapped ADJOINT -- second-argument sums of the one-band rbf adjoint on (y, x) -- exists for Gram calls only)
    no_swap = op == OP_ADJOINT and (not gram or routes.no_adjoint_swap or not hasattr(be, "second_argument_gradient"))
    r = _route_query(be.route, op, fused[0], Xd.shape[2], Xd.shape[1], Yd.shape[1], dyadic, bool(naive), Xd.element_size(), routes.no_stream, no_swap)
    if r in (FUSED_MB, FUSED_MB_SWAP) and routes.no_fused_mb:
        return STREAM
    return r


def _fused_forward(be, static_kernel, Xd, Yd, dyadic, naive, gram, keep_edges=False):
    """Whole forward in one kernel when sk_route_query says so (exactly LinearKernel / RBFKernel, dim <= 16, dyadic <= 2): the
    increments are formed inside the solver.  FUSED: sk_solve_fwd_linear_* / sk_solve_fwd_rbf_*; FUSED_MB (several bands per pair,
    wide paths): sk_solve_fwd_static_*.  None otherwise.  keep_edges (a gradient is pending): (K, edges) -- the edges of the family
    the ADJOINT route names, None when that adjoint forms its own (fp32 paths) or streams."""
    fused = _fused_static(static_kernel, gram)
    if fused is None or not hasattr(be, "route"):      # (the tests' oracle-backed back-end has no fused kernels: it streams)
        return None
    kind, param = fused
    Xd, Yd = Xd.contiguous(), Yd.contiguous()
    one_band = be.solve_fwd_fused_linear if kind == 0 else be.solve_fwd_fused_rbf
    f32 = Xd.dtype == torch.float32
    if keep_edges:
        ra = _route(be, OP_ADJOINT, static_kernel, Xd, Yd, dyadic, naive, gram)
        res = None
        if ra == FUSED and f32:
            # fp32 paths are staged and swept in fp64 whatever their dtype, and the one-band adjoint reads fp64 edges: the forward of
            # the up-cast paths gives the same values (rounded to fp32 below exactly as the fp32-output variant rounds them) AND the
            # edges, so that backward does not sweep forward a second time (Gram + backward 17.2 -> 13.5 ms at 512 x 512 pairs of C4's shape)
            res = one_band(Xd.double(), Yd.double(), param, dyadic, naive, gram, keep_edges=True)
            if res is not None:
                res = (res[0].to(Xd.dtype), res[1])
        elif ra == FUSED:
            res = one_band(Xd, Yd, param, dyadic, naive, gram, keep_edges=True)
        elif ra == FUSED_MB:
            res = be.solve_fwd_fused_static(kind, param, Xd, Yd, dyadic, naive, gram, keep_edges=True)
        elif ra == FUSED_SWAP and gram:
            # long first paths, short second ones: forward AND adjoint on (y, x) -- the edges are those of the pairs (b, a); fp32 paths
            # up-cast as for FUSED above
            res = one_band(Yd.double(), Xd.double(), param, dyadic, naive, True, keep_edges=True) if f32 else \
                one_band(Yd, Xd, param, dyadic, naive, True, keep_edges=True)
            if res is not None:
                res = (res[0].t().contiguous().to(Xd.dtype), res[1])
        if res is not None:
            return res
    rf = _route(be, OP_FORWARD, static_kernel, Xd, Yd, dyadic, naive, gram)
    res = None
    if rf == FUSED and keep_edges:
        # a streaming adjoint ahead: the one-band forward's strip edges are the layout sk_solve_adj_* reads (SK_FLAG_EDGES_GIVEN),
        # so that adjoint skips its own forward sweep
        res = one_band(Xd, Yd, param, dyadic, naive, gram, keep_edges=True)
        if res is not None:
            return res
    if rf not in (FUSED, FUSED_SWAP) and f32 and kind == 1 and dyadic == 0 and not keep_edges and not routes.no_fused_rbf:
        # the one-band RBF kernel at dyadic 0 is built for fp64 paths only: fp32 paths take it up-cast (they are staged in fp64 anyway)
        r8 = _route_query(be.route, OP_FORWARD, kind, Xd.shape[2], Xd.shape[1], Yd.shape[1], dyadic, bool(naive), 8, routes.no_stream)
        if r8 in (FUSED, FUSED_SWAP):
            res = one_band(Xd.double(), Yd.double(), param, dyadic, naive, gram) if r8 == FUSED else \
                one_band(Yd.double(), Xd.double(), param, dyadic, naive, gram)
            if res is not None:
                return (res.t().contiguous() if (r8 == FUSED_SWAP and gram) else res).to(Xd.dtype)
    if rf == FUSED:
        res = one_band(Xd, Yd, param, dyadic, naive, gram)
    elif rf == FUSED_SWAP:
        # long first paths, short second ones: k(y, x) through the one-band kernel (k and both static kernels are symmetric)
        res = one_band(Yd, Xd, param, dyadic, naive, gram)
        if res is not None and gram:
            res = res.t().contiguous()
    elif rf in (FUSED_MB, FUSED_MB_SWAP):
        res = be.solve_fwd_fused_static(kind, param, Xd, Yd, dyadic, naive, gram, swap=rf == FUSED_MB_SWAP)
    if res is None:
        return None
    return (res, None) if keep_edges else res


def _increments(be, static_kernel, Xd, Yd, gram, from_kernel=None):
    """Coarse increments of the static Gram for a tile: fused kernel when available, else the reference's route
    (static kernel in torch -> 4-corner difference, sigkernel.py:216-217 / :362-363).  from_kernel (a list): receives True when the
    fused static kernel formed them (what _tile_gradient's Linear / RBF branch would form again)."""
    fused = _fused_static(static_kernel, gram)
    if fused is not None and hasattr(be, "static_increments"):
        inc = be.static_increments(fused[0], fused[1], Xd.contiguous(), Yd.contiguous(), gram)
        if inc is not None:
            if from_kernel is not None:
                from_kernel.append(True)
            return inc
    G = (static_kernel.Gram_matrix(Xd, Yd) if gram else static_kernel.batch_kernel(Xd, Yd)).contiguous()
    return be.increments(G)


def _mb_pair_bytes(be, kind, Xd, Yd, dyadic):
    """Bytes per pair of what the multi-band fused routes hold in HBM: the edges sk_solve_fwd_static_* keeps (every band's bottom
    row and the terminal column), the adjoint's partial sums and its node-row-0 weights."""
    lay = be._adjoint_mb_layout(1, Xd.shape[1] - 1, Yd.shape[1] - 1, dyadic, Xd.shape[2], kind)
    if lay is None:
        return None
    _, rows, outw, edge_doubles, _, ncols = lay
    return 8 * (edge_doubles + rows * outw + ncols) + 64


def _upcast_tile(X, dyadic):
    return X.dtype == torch.float32 and dyadic == 2


def _tile_gradient(be, static_kernel, Xt, Yt, go, dyadic, naive, gram, edges=None):
    """dL/dX for one tile on the unfused routes: increments -> adjoint PDE (W = dK/d inc_c) -> chain through the static kernel.

    Linear / RBF: sk_static_increments -> sk_solve_adj -> sk_static_adjoint.
    Generic (any duck-typed static kernel): G with autograd -> sk_increments -> sk_solve_adj ->
    sk_increments_adjoint (scaled by the upstream gradient) -> one vector-Jacobian product through the static kernel;
    this replaces the reference's h = 1e-9 finite difference (sigkernel.py:313-341, :472-500)."""
    fused = _fused_static(static_kernel, gram)
    if fused is not None and hasattr(be, "static_adjoint"):
        if _upcast_tile(Xt, dyadic) and edges is None:
            # fp32 paths at dyadic 2: the streaming adjoint exists for fp64 there.  The whole tile in fp64 -- increments formed
            # from the up-cast paths, adjoint, chain rule -- instead of fp32 increments up-cast in 4 GB chunks (round 2: 3x slower)
            g = _tile_gradient(be, static_kernel, Xt.double(), Yt.double(), go.double(), dyadic, naive, gram)
            return g.to(Xt.dtype)
        inc = getattr(edges, "_sk_increments", None) if edges is not None else None      # (what _gram_block kept with the edges)
        if inc is None or inc.shape[:-2] != ((Xt.shape[0], Yt.shape[0]) if gram else (Xt.shape[0],)) or inc.dtype != Xt.dtype:
            inc = be.static_increments(fused[0], fused[1], Xt, Yt, gram)
        if inc is not None:
            _, W = be.solve_adj(inc, dyadic, naive, edges=edges) if edges is not None else be.solve_adj(inc, dyadic, naive)
            del inc
            if getattr(edges, "_sk_increments", None) is not None:
                edges._sk_increments = None      # (freed before the chain rule allocates)
            return be.static_adjoint(fused[0], fused[1], Xt, Yt, W, go, gram)
    Xg = Xt.clone().requires_grad_(True)
    with torch.enable_grad():
        G = static_kernel.Gram_matrix(Xg, Yt) if gram else static_kernel.batch_kernel(Xg, Yt)
    inc = be.increments(G.detach().contiguous())
    _, W = be.solve_adj(inc, dyadic, naive, edges=edges) if edges is not None else be.solve_adj(inc, dyadic, naive)
    del inc
    dG = be.increments_adjoint(W, go)
    del W
    (g,) = torch.autograd.grad(G, Xg, dG)
    return g


def _fused_gradient(be, static_kernel, Xd, Yd, go, dyadic, naive, gram, kept, budget, Kvals=None, route=FUSED):
    """dL/dX for all rows through the fused adjoints -- sk_linear_adjoint_fused_f64 / sk_rbf_adjoint_fused_f64 (route FUSED) or their
    multi-band forms (FUSED_MB): adjoint PDE and the static kernel's chain rule in one kernel, from the paths and the forward's
    edges; no matrix of size pairs x M x N -- one launch per row tile.  None when the kernel does not cover the case after all (the
    caller then takes the streaming route, tiled by ITS transient memory).  Exploding kernels are the library's business: with the
    forward values (Kvals, or what the forward re-run here returns) the launch takes such pairs out of the sweep and adds their
    exact, stored-grid share on the device (csrc/sk_adj_fused_rescue.hip) -- nothing is read back, a backward pass has no host
    synchronisation."""
    A, M = Xd.shape[0], Xd.shape[1]
    kind, param = _fused_static(static_kernel, gram)
    linear = kind == 0
    if route == FUSED_MB:
        adj_mb = be.linear_adjoint_fused_mb if linear else be.rbf_adjoint_fused_mb
        # long / wide paths: sk_solve_fwd_static_* (edges) + the multi-band adjoint; per pair the edges, the partial sums and node row 0
        pair_bytes = _mb_pair_bytes(be, kind, Xd, Yd, dyadic)
        if pair_bytes is None:
            return None
        per_row = _cap_rows((Yd.shape[0] if gram else 1) * pair_bytes, Yd.shape[0] if gram else 1, budget)
        grad = torch.empty_like(Xd)
        for a0, a1, edges in _edge_tiles(kept, A, per_row, budget, strict=True):
            Xt = Xd[a0:a1].contiguous()
            Yt = Yd if gram else Yd[a0:a1].contiguous()
            got = go if go is None else go[a0:a1].reshape(-1).contiguous()
            Kt = None if Kvals is None else Kvals[a0:a1]
            res = adj_mb(Xt, Yt, param, dyadic, edges, got, gram=gram, kfinal=Kt, naive=naive) if edges is not None else None
            if res is None:    # no edges kept, or kept by another kernel in its own layout
                fw = be.solve_fwd_fused_static(kind, param, Xt, Yt, dyadic, naive, gram, keep_edges=True)
                edges = fw[1] if fw is not None else None
                if edges is None:
                    return None
                res = adj_mb(Xt, Yt, param, dyadic, edges, got, gram=gram, kfinal=fw[0], naive=naive)
                del edges
                if res is None:
                    return None
            grad[a0:a1] = res[0]
        return grad
    per_row = _cap_rows((64 * Yd.shape[0] + 2048 * M) if gram else 4096 * M, Yd.shape[0] if gram else 1, budget)      # edges and partial sums only
    grad = torch.empty_like(Xd)
    for a0, a1, edges in _edge_tiles(kept, A, per_row, budget):
        Xt = Xd[a0:a1].contiguous()
        Yt = Yd if gram else Yd[a0:a1].contiguous()
        Kt = None if Kvals is None else Kvals[a0:a1]
        if edges is None:     # (fp32 paths are swept in fp64: edges of the up-cast paths -- which is what their forward kept)
            fwd = be.solve_fwd_fused_linear if linear else be.solve_fwd_fused_rbf
            res = fwd(Xt.double(), Yt.double(), param, dyadic, naive, gram, keep_edges=True)
            edges = res[1] if res is not None else None
            Kt = res[0] if res is not None else None
        if edges is None:
            return None
        adj = be.linear_adjoint_fused if linear else be.rbf_adjoint_fused
        res = adj(Xt, Yt, param, dyadic, edges, None if go is None else go[a0:a1].reshape(-1).contiguous(), gram=gram, kfinal=Kt, naive=naive)
        if res is None:
            return None
        if a0 == 0 and a1 == A and res[0].shape == Xd.shape and res[0].dtype == Xd.dtype:
            return res[0]                # one tile: the kernel's own output, no copy
        grad[a0:a1] = res[0]
    return grad


def _swapped_gradient(be, static_kernel, Xd, Yd, go, dyadic, naive, kept, budget, Kvals=None):
    """dL/dX of a Gram block whose FIRST paths are long and whose second paths fit the one-band adjoints' lanes (route FUSED_SWAP):
    the adjoint runs on the pairs (y_b, x_a) -- k and the static kernel are symmetric -- with the SECOND-argument sums of that sweep
    (sk_rbf_adjoint_fused_f64 / sk_linear_adjoint_fused_f64 with ypart), folded with the transposed upstream gradient:
    d k(x_a, y_b) / d x_a = d2 k(y_b, x_a).  Tiled over the rows of Y by the memory of the sums (48 / 64 bytes per pair and column).
    kept: the edges the forward kept for the swapped pairs, else they are formed here.  None where the kernel declines (the caller streams)."""
    A, B = Xd.shape[0], Yd.shape[0]
    kind, param = _fused_static(static_kernel, True)
    linear = kind == 0
    fwd, adj = (be.solve_fwd_fused_linear, be.linear_adjoint_fused) if linear else (be.solve_fwd_fused_rbf, be.rbf_adjoint_fused)
    edges = kept[0][2] if (kept and len(kept) == 1 and kept[0][:2] == (0, A) and kept[0][2] is not None) else None
    KT = None if Kvals is None else Kvals.t().contiguous()
    if edges is None:
        res = fwd(Yd, Xd, param, dyadic, naive, True, keep_edges=True)
        if res is None or res[1] is None:
            return None
        KT, edges = res
    per = edges.numel() // B
    goT = go.t().contiguous()                     # upstream gradient of the pair (b, a)
    grad = None
    for b0, b1 in _tiles(B, 64 * A * (Xd.shape[1] + 16), budget):
        res = adj(Yd[b0:b1].contiguous(), Xd, param, dyadic, edges[b0 * per:b1 * per], None, gram=True, yside=True,
                  kfinal=None if KT is None else KT[b0:b1], naive=naive, **({} if linear else {"yonly": True}))
        if res is None:
            return None
        g = be.second_argument_gradient(res[2], Xd, None if linear else param, goT[b0:b1], 0)
        grad = g if grad is None else grad + g
        del res
    return grad


def _rows_gradient(be, static_kernel, Xd, Yd, go, dyadic, naive, gram, kept, workspace_bytes, Kvals=None):
    """dL/dX (A,M,D) of a Gram block (gram=True: go (A,B)) or a paired batch (go (A,)): the fused linear / RBF adjoint when it
    applies, else the unfused routes tiled over rows by their transient memory (3 (Linear/RBF) or 8 (generic) arrays of the
    size of the tile's increments).  kept: what forward left for the tiles ([(a0, a1, edges)] or None)."""
    A, M, N = Xd.shape[0], Xd.shape[1], Yd.shape[1]
    budget = _budget(Xd.device, workspace_bytes)
    route = _route(be, OP_ADJOINT, static_kernel, Xd, Yd, dyadic, naive, gram)
    if route in (FUSED, FUSED_MB):
        g = _fused_gradient(be, static_kernel, Xd, Yd, go, dyadic, naive, gram, kept, budget, Kvals, route)
        if g is not None:
            return g
    if route == FUSED_SWAP and gram:
        # (fp32 paths: swept in fp64 like every one-band call -- the edges their forward kept are those of the up-cast paths)
        g = _swapped_gradient(be, static_kernel, Xd.double(), Yd.double(), go.double(), dyadic, naive, kept, budget,
                              None if Kvals is None else Kvals.double())
        if g is not None:
            return g.to(Xd.dtype)
        kept = None       # (edges of the swapped pairs are of no use to the streaming route below)
    fused = _fused_static(static_kernel, gram) is not None
    esize = 8 if (fused and _upcast_tile(Xd, dyadic)) else Xd.element_size()
    per_row = (3 if fused else 8) * (Yd.shape[0] if gram else 1) * M * N * esize
    grad = torch.empty_like(Xd)
    for a0, a1, edges in _edge_tiles(kept, A, per_row, budget, strict=True):
        grad[a0:a1] = _tile_gradient(be, static_kernel, Xd[a0:a1].contiguous(), Yd if gram else Yd[a0:a1].contiguous(),
                                     go[a0:a1].contiguous(), dyadic, naive, gram, edges=edges)
    return grad


def _check_inputs(X, Y, paired):
    if X.dim() != 3 or Y.dim() != 3:
        raise ValueError("X and Y must have shape (batch, length, dim)")
    if X.shape[2] != Y.shape[2]:
        raise ValueError("X and Y must have the same path dimension")
    if paired and X.shape[0] != Y.shape[0]:
        raise ValueError("compute_kernel needs the same batch size for X and Y")
    if X.dtype != Y.dtype or X.device != Y.device:
        raise ValueError("X and Y must share dtype and device")


class _SigKernel(torch.autograd.Function):
    """k_sig(x_i, y_i) for paired batches -- the reference's ``_SigKernel`` (sigkernel.py:201-343)."""

    @staticmethod
    def forward(ctx, X, Y, static_kernel, dyadic_order, _naive_solver=False, workspace_bytes=None):
        _check_inputs(X, Y, paired=True)
        be = _lib.get_backend()
        A, M, N = X.shape[0], X.shape[1], Y.shape[1]
        ctx.save_for_backward(X, Y)
        ctx.static_kernel, ctx.dyadic_order, ctx._naive_solver = static_kernel, dyadic_order, _naive_solver
        ctx.workspace_bytes = workspace_bytes
        if M < 2 or N < 2 or A == 0:  # a single point: the grid is its boundary, k = 1 (sigkernel.py:212-253 with MM = 0);
            return torch.ones(A, dtype=X.dtype, device=X.device)   # an empty batch: an empty result, like the CPU reference
        Xd, Yd = X.detach(), Y.detach()
        ctx.kept_edges = None
        # (whether a gradient can be ASKED for -- not X.requires_grad: under torch.no_grad() a leaf X must not make the forward keep edges,
        # take the adjoint's kernel family or up-cast fp32 paths)
        need = bool(ctx.needs_input_grad[0])
        if need and hasattr(be, "solve_fwd_keep_edges"):
            # a gradient is pending: the forward keeps the terminal edges of every pair (8 (MM + NN) bytes each; a paired batch is
            # small), so that backward is ONE adjoint launch instead of a second forward sweep + the adjoint
            edge_bytes = 8.0 * A * (((M - 1) << dyadic_order) + ((N - 1) << dyadic_order) + 32)
            if _route(be, OP_ADJOINT, static_kernel, Xd, Yd, dyadic_order, _naive_solver, False) == FUSED_MB:
                pair_bytes = _mb_pair_bytes(be, _fused_static(static_kernel, False)[0], Xd, Yd, dyadic_order)
                edge_bytes = float(A) * (pair_bytes or 0)
            if edge_bytes <= _cost("keep_edges_fraction") * _budget(X.device, workspace_bytes):
                res = _fused_forward(be, static_kernel, Xd, Yd, dyadic_order, _naive_solver, gram=False, keep_edges=True)
                if res is not None:
                    K, edges = res
                    if edges is not None:
                        ctx.kept_edges = [(0, A, edges)]
                    ctx.K = K.detach()
                    return K
        K = _fused_forward(be, static_kernel, Xd, Yd, dyadic_order, _naive_solver, gram=False)
        if K is not None:
            ctx.K = K.detach() if need else None     # forward values: what arms the fused adjoint's device-side rescue
            return K
        K = torch.empty(A, dtype=X.dtype, device=X.device)
        per_row = 2 * M * N * X.element_size()
        for a0, a1 in _tiles(A, per_row, _budget(X.device, workspace_bytes)):
            inc = _increments(be, static_kernel, Xd[a0:a1], Yd[a0:a1], gram=False)   # sigkernel.py:216-217 (:218 by index)
            K[a0:a1] = be.solve_fwd(inc, dyadic_order, _naive_solver)               # :231 / :246
        return K

    @staticmethod
    def backward(ctx, grad_output):
        X, Y = ctx.saved_tensors
        sk, d, naive = ctx.static_kernel, ctx.dyadic_order, ctx._naive_solver
        be = _lib.get_backend()
        A, M, N = X.shape[0], X.shape[1], Y.shape[1]
        if M >= 2 and N >= 2 and A > 0:
            go = grad_output.to(X.dtype).contiguous()
            kept, ctx.kept_edges = getattr(ctx, "kept_edges", None), None
            grad_X = _rows_gradient(be, sk, X.detach().contiguous(), Y.detach().contiguous(), go, d, naive, False, kept,
                                    ctx.workspace_bytes, getattr(ctx, "K", None))
        else:
            grad_X = torch.zeros_like(X)     # single points / an empty batch: k = 1 whatever X is
        return grad_X, None, None, None, None, None


# Cost rules (WHEN a route is the faster one -- not scope rules) live in ONE table, the library's (sk_cost_query, csrc/sk_route.hip: value +
# the A/B measurement behind it; tools/crossovers.py re-measures them).  None = the table's value; tests set an attribute to override.
_SYM_TILES = None              # "sym_tiles": row tiles of the symmetric shortcut: work = (T + 1) / (2 T) of the full Gram
_SYM_MIN_CELLS = None          # "sym_min_cells": below this the extra launches cost more than the saved solves
_SYM_MIN_ROWS = None           # "sym_min_rows": rows per block of the triangular adjoint
_SYM_STREAM_MIN_PATHS = None   # "sym_stream_min_paths": the streaming route's symmetric forward takes the blocked triangle from this many paths
_KEEP_EDGES_FRACTION = None    # "keep_edges_fraction": of the transient budget, what may stay allocated between forward and backward
_KEEP_INCREMENTS_FRACTION = None   # "keep_increments_fraction": ... and the streaming route's increments of a one-tile block beside them
_PAIRED_MERGE_CELLS = None     # "paired_merge_cells"
_MMD_STREAMS_MAX_PAIRS = None  # "mmd_streams_max_pairs"


def _cost(name):
    v = globals()["_" + name.upper()]
    return _lib.cost(name) if v is None else v


_MAX_LAUNCH_PAIRS = 1 << 30  # pairs per fused launch (the fused kernels index pairs with 32 bits and refuse 2^31 - 2^20 and more)


def _gram_block(be, static_kernel, Xd, Yd, dyadic_order, naive, workspace_bytes, rows_factor=None, keep=None):
    """K[a, b] for every pair of Xd x Yd (no autograd): fused kernel, or increments + solver tiled over rows of Xd.

    keep (a list, when a gradient is pending): receives one (a0, a1, edges) per tile -- the terminal row/column of every
    pair, 8(MM+NN) bytes per pair, which lets backward skip its forward sweep.  The reference keeps the whole solution
    grid for the same purpose (sigkernel.py:248, :397-399)."""
    A, B, M, N = Xd.shape[0], Yd.shape[0], Xd.shape[1], Yd.shape[1]
    if A * B > _MAX_LAUNCH_PAIRS and A > 1:
        # more pairs than one fused launch indexes (32 bits): row tiles of at most _MAX_LAUNCH_PAIRS pairs, each on its own route
        # (46400 x 46400 paths of 16 points, tools/experiments/r04_huge_batch.py: 2.8 s and 26 GB at the peak in three fused tiles against
        # 5.2 s and 114 GB streamed)
        rows = max(1, _MAX_LAUNCH_PAIRS // B)
        K = torch.empty(A, B, dtype=Xd.dtype, device=Xd.device)
        kept_all = []
        for a0 in range(0, A, rows):
            a1 = min(a0 + rows, A)
            sub = [] if keep is not None else None
            K[a0:a1] = _gram_block(be, static_kernel, Xd[a0:a1], Yd, dyadic_order, naive, workspace_bytes, rows_factor, sub)
            if sub:
                kept_all.extend((a0 + s0, a0 + s1, e) for s0, s1, e in sub)
        # edges only when every tile kept them for all of its rows (else backward tiles by its own budget and sweeps forward itself)
        if keep is not None and kept_all and kept_all[0][0] == 0 and kept_all[-1][1] == A \
                and all(kept_all[i][1] == kept_all[i + 1][0] for i in range(len(kept_all) - 1)):
            keep.extend(kept_all)
        return K
    budget = None   # (asked of the device only where it is needed: a small fused call does not pay for hipMemGetInfo)
    if keep is not None and hasattr(be, "solve_fwd_keep_edges"):
        budget = _budget(Xd.device, workspace_bytes)
        edge_bytes = 8.0 * A * B * (((M - 1) << dyadic_order) + ((N - 1) << dyadic_order) + 32)
        if _route(be, OP_ADJOINT, static_kernel, Xd, Yd, dyadic_order, naive, True) == FUSED_MB:
            # the multi-band forward keeps every band's bottom row: nb / 2 times the terminal row and column alone
            pair_bytes = _mb_pair_bytes(be, _fused_static(static_kernel, True)[0], Xd, Yd, dyadic_order)
            edge_bytes = float(A) * B * (pair_bytes or 0)
        if edge_bytes > _cost("keep_edges_fraction") * budget:
            keep = None
    else:
        keep = None
    if keep is not None:
        res = _fused_forward(be, static_kernel, Xd, Yd, dyadic_order, naive, gram=True, keep_edges=True)
        if res is not None:
            K, edges = res
            if edges is not None:
                keep.append((0, A, edges))      # one block for all rows: backward slices it per tile
            return K
    else:
        K = _fused_forward(be, static_kernel, Xd, Yd, dyadic_order, naive, gram=True)
        if K is not None:
            return K
    K = torch.empty(A, B, dtype=Xd.dtype, device=Xd.device)
    fused = _fused_static(static_kernel, True) is not None
    if budget is None:
        budget = _budget(Xd.device, workspace_bytes)
    # transient bytes per Gram row: G_static + inc_c on the generic route, inc_c alone on the fused one
    per_row = (rows_factor or (1 if fused else 2)) * B * M * N * Xd.element_size()
    tiles = _tiles(A, per_row, budget)
    for a0, a1 in tiles:
        by_kernel = []
        inc = _increments(be, static_kernel, Xd[a0:a1], Yd, True, by_kernel)     # sigkernel.py:362-363 (:364 by index)
        if keep is not None:
            K[a0:a1], edges = be.solve_fwd_keep_edges(inc, dyadic_order, naive)  # :378 / :395, + the edges for backward
            if edges is not None and by_kernel and len(tiles) == 1 and \
                    inc.numel() * inc.element_size() <= _cost("keep_increments_fraction") * budget:
                # one tile of moderate size: its increments ride along with the edges, and backward does not evaluate the static
                # kernel a second time (a quarter of a gradient step on wide paths, profiles/r06_keep_inc.txt)
                edges._sk_increments = inc
            keep.append((a0, a1, edges))
        else:
            K[a0:a1] = be.solve_fwd(inc, dyadic_order, naive)                    # :378 / :395
    return K


def _gram_symmetric(be, static_kernel, Xd, dyadic_order, naive, workspace_bytes, keep_blocks=None):
    """compute_Gram(X, X, sym=True): only the blocks on and above the diagonal of a T x T tiling are solved and
    mirrored, like the reference's CPU solver does pair by pair (cython_backend.pyx:74-97; its GPU path ignores `sym`).
    The result is exactly symmetric.  keep_blocks (a list, when a gradient is pending) receives (r0, r1, kept edges) per
    row block for _SigKernelGram.backward's triangular adjoint."""
    A = Xd.shape[0]
    if keep_blocks is None and hasattr(be, "solve_fwd_fused_sym"):
        # exactly LinearKernel / RBFKernel within the single-band fused kernels' scope: the triangle in ONE launch, every value
        # written to both halves (sk_solve_fwd_linear_sym_* / sk_solve_fwd_rbf_sym_*)
        if _route(be, OP_FORWARD, static_kernel, Xd, Xd, dyadic_order, naive, True) == FUSED:
            kind, param = _fused_static(static_kernel, True)
            K = be.solve_fwd_fused_sym(kind, param, Xd, dyadic_order, naive)
            if K is not None:
                return K
    K = torch.empty(A, A, dtype=Xd.dtype, device=Xd.device)
    cells = float(A) * A * ((Xd.shape[1] - 1) << dyadic_order) ** 2
    # 8 block launches instead of 1: only worth it when the solve dwarfs the launches (measured: 128 x 128 pairs of
    # length 64 take 0.4 ms in one launch, 0.8 ms in blocks)
    tiles = int(_cost("sym_tiles"))
    T = tiles if (A >= 8 * tiles and cells >= _cost("sym_min_cells")) else 1
    if T == 1 and keep_blocks is None and A >= _cost("sym_stream_min_paths") and _SYM_MIN_CELLS is None and \
            _route(be, OP_FORWARD, static_kernel, Xd, Xd, dyadic_order, naive, True) == STREAM:
        # the streaming route (wide paths, dyadic >= 3, user-defined kernels): the static kernel and the increments are most of a pair's
        # cost, the blocked triangle pays from ~200 paths on whatever the grid (profiles/r06_sym_stream.txt)
        T = tiles
    if T > 1 and A >= 64 * T and keep_blocks is not None:
        T *= 2        # big batches with the fused adjoint: 16 row blocks solve 53 % of the square instead of 56 % (C4: -1 %)
    if keep_blocks is not None and T > 1:
        # with the adjoint in the blocks too, small blocks lose more to launches and pipeline fill than the triangle saves
        # (measured: 64 paths of length 700 in 8 blocks of 8 rows: backward 60 -> 74 ms): at least _SYM_MIN_ROWS rows each
        T = max(1, min(T, A // int(_cost("sym_min_rows"))))
    step = -(-A // T)
    for r0 in range(0, A, step):
        r1 = min(r0 + step, A)
        kept = [] if keep_blocks is not None else None
        blk = _gram_block(be, static_kernel, Xd[r0:r1].contiguous(), Xd[r0:].contiguous(), dyadic_order, naive, workspace_bytes,
                          3 if keep_blocks is not None else None, kept)
        if keep_blocks is not None:
            keep_blocks.append((r0, r1, kept))
        K[r0:r1, r0:] = blk
        if r1 < A:
            K[r1:, r0:r1] = blk[:, r1 - r0:].t()
    # the diagonal blocks were solved in full: symmetrise them to the upper triangle too
    iu = torch.triu_indices(A, A, offset=1, device=Xd.device)
    K[iu[1], iu[0]] = K[iu[0], iu[1]]
    return K


def _edge_tiles(kept, n_rows, per_row, budget, strict=False):
    """Row tiles of a backward pass with the terminal edges forward kept for them: [(a0, a1, edges or None)].
    strict: never hand back a tile with more rows than `budget` allows (the kept tiling may have been sized for a route with
    less transient memory); kept edges that do not match the tiling are then dropped and the adjoint sweeps forward itself."""
    tiles = [(a0, a1, None) for a0, a1 in _tiles(n_rows, per_row, budget)]
    if kept and len(kept) == 1 and kept[0][:2] == (0, n_rows) and kept[0][2] is not None:
        if len(tiles) == 1:
            return kept
        full = kept[0][2]              # the fused forward kept one block for all rows: slice it per tile
        per = full.numel() // n_rows
        return [(a0, a1, full[a0 * per:a1 * per]) for a0, a1, _ in tiles]
    if kept and (not strict or max(a1 - a0 for a0, a1, _ in kept) <= tiles[0][1] - tiles[0][0]):
        return kept                    # the tiling of forward, with the edges it kept (None where the strip kernels did not apply)
    return tiles


def _sym_triangle_ok(be, static_kernel, Xd, dyadic, naive):
    """Whether compute_Gram(X, X, sym=True) WITH a gradient solves the triangle only (a pair above the diagonal also stands for its
    mirror image, through the second-argument contraction of the same adjoint sweep), by the ADJOINT route of the shape:
      FUSED, RBFKernel, fp64   yes where sk_route_query(SK_OP_ADJOINT_SYM) says so -- sk_rbf_adjoint_fused_f64 with the second-argument sums
                               (_sym_fused_gradient; dim <= 4, 64 points at dyadic 1..2, 128 at dyadic 0); all pairs beyond
      FUSED, LinearKernel      no  -- the fused adjoint on ALL pairs is faster than any triangle route (14 vs 20 ms at the C3 shape)
      FUSED_MB                 no  -- likewise (C5's shape: 0.29 s on all pairs against 0.46 s on the streamed triangle)
      STREAM (fused static kernels beyond dim 16 / dyadic 2, or a route switch): yes where sk_static_adjoint2 exists (linear: dim <= 8)
    One predicate for the single-GPU Function and the sharded one (sigkernel_amd.distributed)."""
    fused = _fused_static(static_kernel, True)
    if fused is None or not hasattr(be, "static_adjoint2"):
        return False
    route = _route(be, OP_ADJOINT, static_kernel, Xd, Xd, dyadic, naive, True)
    if route == FUSED:      # (the library says where the second-argument sums exist and pay: sk_route_query(SK_OP_ADJOINT_SYM))
        return hasattr(be, "second_argument_gradient") and _route(be, OP_ADJOINT_SYM, static_kernel, Xd, Xd, dyadic, naive, True) == FUSED
    if route == STREAM:
        return Xd.shape[2] <= (8 if fused[0] == 0 else 32)
    return False


def _same_storage(Xd, Yd):
    return Xd.shape == Yd.shape and Xd.data_ptr() == Yd.data_ptr() and Xd.stride() == Yd.stride()


def _sym_fused_gradient(be, static_kernel, Xd, go, dyadic, naive, sym_blocks, budget, Kvals=None):
    """dL/dX of compute_Gram(X, X, sym=True) from the triangular row blocks through the FUSED RBF adjoint with the
    second-argument sums (sk_rbf_adjoint_fused_f64 with ypart): per row block r0:r1 ONE launch over the solved pairs
    (a in r0:r1, b >= r0) gives the first-argument rows r0:r1 and, per pair, the sums that what the unsolved mirror pairs
    (b, a), b >= r1, owe to rows r1: is folded from (d1 K(x_b, x_a) = d2 K(x_a, x_b): the scheme is symmetric).  Neither the
    increments nor W exist in HBM, and exploding pairs are rescued on the device (Kvals: the forward's (A, A) values).  None when
    the kernel does not cover the case or the forward kept no edges; the caller then takes the unfused triangular route."""
    if not (type(static_kernel) is RBFKernel and _route(be, OP_ADJOINT, static_kernel, Xd, Xd, dyadic, naive, True) == FUSED
            and Xd.dtype == torch.float64 and hasattr(be, "second_argument_gradient")):
        return None
    A, M = Xd.shape[0], Xd.shape[1]
    sigma = float(static_kernel.sigma)
    grad = torch.zeros_like(Xd)
    for r0, r1, kept in sym_blocks:
        if not kept or len(kept) != 1 or kept[0][2] is None or kept[0][:2] != (0, r1 - r0):
            return None
        Xc = Xd[r0:].contiguous()
        nb = Xc.shape[0]
        # rows per launch by the memory of the second-argument sums (48 bytes per pair and node column)
        per_row = 64 * nb * (M + 16)
        edges = kept[0][2]
        per = edges.numel() // (r1 - r0)
        for a0, a1 in _tiles(r1 - r0, per_row, budget):
            Xt = Xd[r0 + a0:r0 + a1].contiguous()
            Kt = None if Kvals is None else Kvals[r0 + a0:r0 + a1, r0:]
            res = be.rbf_adjoint_fused(Xt, Xc, sigma, dyadic, edges[a0 * per:a1 * per], go[r0 + a0:r0 + a1, r0:].reshape(-1).contiguous(),
                                       gram=True, yside=r1 < A, kfinal=Kt, naive=naive)
            if res is None:
                return None
            grad[r0 + a0:r0 + a1] += res[0]
            if r1 < A:   # upstream gradient of the mirror pair (b, a) is go[b, a]
                grad[r1:] += be.second_argument_gradient(res[2], Xc, sigma, go[r0:, r0 + a0:r0 + a1].t(), r1 - r0)
            del res
    return grad


def _sym_unfused_gradient(be, kind, param, Xd, go, dyadic, naive, sym_blocks, budget):
    """The same through sk_static_increments -> sk_solve_adj -> sk_static_adjoint (first argument) + sk_static_adjoint2 (second
    argument of the SAME W, weighted by the transposed upstream gradient); owns the stored-grid rescue."""
    A, M, N = Xd.shape[0], Xd.shape[1], Xd.shape[1]
    if _upcast_tile(Xd, dyadic) and all(not kept or all(e[2] is None for e in kept) for _, _, kept in sym_blocks):
        # fp32 paths at dyadic 2 without kept edges: the whole route in fp64 (see _tile_gradient)
        return _sym_unfused_gradient(be, kind, param, Xd.double(), go.double(), dyadic, naive, sym_blocks, budget).to(Xd.dtype)
    grad_X = torch.zeros_like(Xd)
    for r0, r1, kept in sym_blocks:
        Xr, Xc = Xd[r0:r1].contiguous(), Xd[r0:].contiguous()
        go_blk = go[r0:r1, r0:].contiguous()
        go_t = go[r0:, r0:r1].t().contiguous()         # [a, b] -> upstream gradient of the mirror pair (b, a)
        per_row = 3 * Xc.shape[0] * M * N * Xd.element_size()
        for a0, a1, edges in _edge_tiles(kept, r1 - r0, per_row, budget):
            Xt = Xr[a0:a1].contiguous()
            inc = getattr(edges, "_sk_increments", None) if edges is not None else None      # (what _gram_block kept with the edges)
            if inc is None or inc.shape[:-2] != (Xt.shape[0], Xc.shape[0]) or inc.dtype != Xt.dtype:
                inc = be.static_increments(kind, param, Xt, Xc, True)
            _, W = be.solve_adj(inc, dyadic, naive, edges=edges) if edges is not None else be.solve_adj(inc, dyadic, naive)
            del inc
            if getattr(edges, "_sk_increments", None) is not None:
                edges._sk_increments = None
            grad_X[r0 + a0:r0 + a1] += be.static_adjoint(kind, param, Xt, Xc, W, go_blk[a0:a1].contiguous(), True)
            if r1 < A:
                g2 = be.static_adjoint2(kind, param, Xt, Xc, W, go_t[a0:a1].contiguous(), r1 - r0)
                if g2 is None:      # (_sym_triangle_ok keeps such shapes off the triangle: a caller that bypassed it)
                    raise RuntimeError("sigkernel_amd: the triangular adjoint has no second-argument kernel for path dim %d" % Xd.shape[2])
                grad_X[r1:] += g2
            del W
    return grad_X


class _SigKernelGram(torch.autograd.Function):
    """Gram matrix k_sig(x_i, y_j) -- the reference's ``_SigKernelGram`` (sigkernel.py:347-416)."""

    @staticmethod
    def forward(ctx, X, Y, static_kernel, dyadic_order, sym=False, _naive_solver=False, workspace_bytes=None):
        _check_inputs(X, Y, paired=False)
        be = _lib.get_backend()
        A, B, M, N = X.shape[0], Y.shape[0], X.shape[1], Y.shape[1]
        ctx.save_for_backward(X, Y)
        ctx.static_kernel, ctx.dyadic_order, ctx._naive_solver = static_kernel, dyadic_order, _naive_solver
        ctx.workspace_bytes = workspace_bytes
        ctx.sym_blocks = ctx.kept_edges = ctx.K = None
        if M < 2 or N < 2 or A == 0 or B == 0:   # single points: k = 1; an empty batch: an empty matrix
            return torch.ones(A, B, dtype=X.dtype, device=X.device)
        Xd, Yd = X.detach(), Y.detach()
        need, need_y = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])      # (not need: see _SigKernel.forward)
        # `sym`: the reference's GPU path ignores it (sigkernel.py:366-382) and its CPU path silently assumes X is Y.
        # Here it halves the work when that assumption can be checked (same storage) and no gradient is needed.
        ctx.sym_blocks = None
        if sym and _same_storage(Xd, Yd):
            if not need and not need_y:
                return _gram_symmetric(be, static_kernel, Xd.contiguous(), dyadic_order, _naive_solver, workspace_bytes)
            # with a gradient: the triangular forward AND a triangular adjoint (a pair above the diagonal also stands for
            # its mirror image, through the second-argument contraction of the same W) -- fused static kernels only
            # (the fused linear adjoint is faster on all pairs than the unfused one on the triangle: 14 vs 20 ms at the C3 shape)
            # (long / wide RBF paths: the multi-band fused adjoint on ALL pairs beats the unfused triangle -- C5's shape 0.29 s against 0.46 s)
            if need and need_y and _sym_triangle_ok(be, static_kernel, Xd, dyadic_order, _naive_solver):
                ctx.sym_blocks = []
                K = _gram_symmetric(be, static_kernel, Xd.contiguous(), dyadic_order, _naive_solver, workspace_bytes,
                                    ctx.sym_blocks)
                ctx.K = K.detach()
                return K
        fused = _fused_static(static_kernel, True) is not None
        # with a gradient pending, tile like backward will, so that the caching allocator can reuse the same blocks
        rows_factor = (3 if fused else 8) if need else None
        ctx.kept_edges = [] if need else None
        K = _gram_block(be, static_kernel, Xd, Yd, dyadic_order, _naive_solver, workspace_bytes, rows_factor, ctx.kept_edges)
        if sym and _same_storage(Xd, Yd):   # all pairs were solved (fused adjoint ahead): still hand back an exactly symmetric matrix
            iu = torch.triu_indices(A, A, offset=1, device=K.device)
            K[iu[1], iu[0]] = K[iu[0], iu[1]]
        ctx.K = K.detach() if need else None     # forward values: what arms the fused adjoints' device-side rescue
        return K

    @staticmethod
    def backward(ctx, grad_output):
        X, Y = ctx.saved_tensors
        sk, d, naive = ctx.static_kernel, ctx.dyadic_order, ctx._naive_solver
        be = _lib.get_backend()
        A, B, M, N = X.shape[0], Y.shape[0], X.shape[1], Y.shape[1]
        grad_X = None
        if M >= 2 and N >= 2 and getattr(ctx, "sym_blocks", None):
            # compute_Gram(X, X, sym=True): per row block r0:r1 the pairs (a, b >= r0) were solved.  Their first-argument
            # contraction gives rows r0:r1; the second-argument contraction of the SAME W, weighted by the transposed
            # upstream gradient, gives what the unsolved mirror pairs (b, a), b >= r1, owe to rows r1: (K is symmetric, so
            # d1 K(x_b, x_a) = d2 K(x_a, x_b)).  The reference's 2x rule (sigkernel.py:410-412) is applied below as usual.
            Xd = X.detach().contiguous()
            go = grad_output.to(X.dtype).contiguous()
            kind, param = _fused_static(sk, True)
            budget = _budget(X.device, ctx.workspace_bytes)
            g_fused = _sym_fused_gradient(be, sk, Xd, go, d, naive, ctx.sym_blocks, budget, getattr(ctx, "K", None))
            if g_fused is not None:
                grad_X = g_fused
            else:
                grad_X = _sym_unfused_gradient(be, kind, param, Xd, go, d, naive, ctx.sym_blocks, budget)
            ctx.sym_blocks = None
        elif M >= 2 and N >= 2 and A > 0 and B > 0:
            go = grad_output.to(X.dtype).contiguous()
            kept, ctx.kept_edges = getattr(ctx, "kept_edges", None), None
            grad_X = _rows_gradient(be, sk, X.detach().contiguous(), Y.detach().contiguous(), go, d, naive, True, kept,
                                    ctx.workspace_bytes, getattr(ctx, "K", None))
        # the reference doubles the gradient when Y requires grad (written for compute_Gram(X, X) with a
        # symmetric grad_output, sigkernel.py:410-412) and never returns a gradient for Y
        if grad_X is None:
            grad_X = torch.zeros_like(X)     # single points / an empty batch: k = 1 whatever X is
        elif ctx.needs_input_grad[1]:
            grad_X = 2 * grad_X
        return grad_X, None, None, None, None, None, None


def k_kgrad(X, Y, gamma, dyadic_order, static_kernel, eps=1e-4, workspace_bytes=None):
    """Signature kernel and its first / second directional derivative along gamma -- the reference's ``k_kgrad``
    (sigkernel.py:504-593): X (A,M,D), Y (B,N,D), gamma (A,M,D) -> three (A,B) matrices
    k(x_a, y_b), d/ds k(x_a + s gamma_a, y_b)|_0, d2/ds2 k(x_a + s gamma_a, y_b)|_0.

    Like the reference, the derivatives of the static kernel are one-sided finite differences with step ``eps``
    (sigkernel.py:529-541; their truncation error is part of the reference's result, so it is reproduced, not
    "fixed"); the three increment arrays then drive ONE sweep of the coupled PDE stencil (cuda_backend.py:206-220)
    in sk_solve_deriv_*.  No autograd: the reference's outputs carry none either (solution buffers are fresh
    tensors, sigkernel.py:553-566)."""
    _check_inputs(X, Y, paired=False)
    if gamma.shape != X.shape or gamma.dtype != X.dtype or gamma.device != X.device:
        raise ValueError("gamma must have X's shape, dtype and device")
    be = _lib.get_backend()
    A, B, M, N = X.shape[0], Y.shape[0], X.shape[1], Y.shape[1]
    out = torch.zeros(3, A, B, dtype=X.dtype, device=X.device)
    if M < 2 or N < 2:
        out[0] = 1.
        return out[0], out[1], out[2]
    Xd, Yd, gd = X.detach(), Y.detach(), gamma.detach()
    fused = _fused_static(static_kernel, True) if hasattr(be, "static_deriv_increments") else None
    # transient bytes per row of X: three increment arrays (+ three static Gram matrices on the generic route)
    per_row = (3 if fused is not None else 6) * B * M * N * X.element_size()
    for a0, a1 in _tiles(A, per_row, _budget(X.device, workspace_bytes)):
        Xt, gt = Xd[a0:a1], gd[a0:a1]
        X1, X2 = Xt + eps * gt, Xt + 2. * eps * gt                                   # sigkernel.py:530, :537
        if fused is not None and hasattr(be, "solve_deriv_fused") and not routes.no_fused_deriv:
            # static kernel, finite differences, increments AND the three-state sweep in one kernel (sk_solve_deriv_static_f64)
            res = be.solve_deriv_fused(fused[0], fused[1], Xt.contiguous(), X1, X2, Yd.contiguous(), dyadic_order, eps)
            if res is not None:
                out[0, a0:a1], out[1, a0:a1], out[2, a0:a1] = res
                continue
        inc3 = None
        if fused is not None:    # static kernel + finite differences + increments in one pass (sk_static_deriv_increments_*)
            inc3 = be.static_deriv_increments(fused[0], fused[1], Xt.contiguous(), X1, X2, Yd.contiguous(), eps)
        if inc3 is None:
            G0 = static_kernel.Gram_matrix(Xt, Yd).contiguous()                      # :526
            G1 = static_kernel.Gram_matrix(X1, Yd).contiguous()                      # :530
            G2 = static_kernel.Gram_matrix(X2, Yd).contiguous()                      # :537
            inc3 = be.deriv_increments(G0, G1, G2, eps)                              # :527-541
            del G0, G1, G2
        k, kd, kdd = be.solve_deriv(inc3, dyadic_order)                              # :543-566 (tile() by index)
        out[0, a0:a1], out[1, a0:a1], out[2, a0:a1] = k, kd, kdd
    return out[0], out[1], out[2]


_SIDE_STREAMS = {}
_LOSS_WEIGHTS = {}


def _side_streams(device):
    """Two side streams per device, created once."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _SIDE_STREAMS[key]


def routes_allow_streams():
    return not routes.no_mmd_streams


def _loss_weights(A, B, dtype, device, max_cached=16):
    """Constant weight matrices of the loss wrappers, built once per (A, B, dtype, device):
       wf (A, A+B)  value:    [ (1 - I) / (A (A-1)) | -2 / (A B) ]   -- K_XX_m - 2 mean(K_XY) = sum(K(X, [X; Y]) * wf)
       wb (A, A+B)  gradient: [ 2 (1 - I) / (A (A-1)) | -2 / (A B) ] -- d loss / dK with the reference's 2x rule on the K_XX part
                              (_SigKernelGram.backward doubles when both arguments require grad, sigkernel.py:410-412)
       wy (B, B)    value:    (1 - I) / (B (B-1))                     -- K_YY_m = sum(K_YY * wy); None for B < 2"""
    key = (A, B, dtype, device)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    ent = _LOSS_WEIGHTS.get(key)
    if ent is not None:
        # a captured hipGraph holds the cached tensors BY ADDRESS: an entry a capture has used is pinned for good (an eviction would
        # let the allocator hand its blocks out again under an already captured training step); everything else is least-recently-used
        _LOSS_WEIGHTS[key] = _LOSS_WEIGHTS.pop(key)[:3] + (ent[3] or capturing,)
        return ent[:3]
    # (built inside a hipGraph capture -- a capture without a warm-up call -- the fills are nodes of that graph and the memory
    # belongs to its pool: such weights serve the captured call only and are not cached)
    wf = torch.empty(A, A + B, dtype=dtype, device=device)
    wf[:, :A] = (1.0 - torch.eye(A, dtype=dtype, device=device)) / (A * (A - 1.0))
    wf[:, A:] = -2.0 / (A * float(B))
    wb = wf.clone()
    wb[:, :A] *= 2.0
    wy = (1.0 - torch.eye(B, dtype=dtype, device=device)) / (B * (B - 1.0)) if B > 1 else None
    w = (wf, wb, wy)
    if not capturing:
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()    # once per shape: later calls may read them from any stream
        _LOSS_WEIGHTS[key] = w + (False,)
        # bounded: workloads whose batch sizes vary would otherwise leak A (A + B) + B^2 elements per distinct shape; the oldest
        # UNPINNED entries go (dicts keep insertion order; a hit re-inserts its entry at the end)
        # (max_cached: weight sets kept besides those a graph capture has pinned)
        if len(_LOSS_WEIGHTS) > max_cached:
            for k in [k for k, e in _LOSS_WEIGHTS.items() if not e[3]][:len(_LOSS_WEIGHTS) - max_cached]:
                del _LOSS_WEIGHTS[k]
    return w


class _SigKernelLoss(torch.autograd.Function):
    """K_XX_m - 2 mean(K_XY) [+ K_YY_m] for TRAINING-SIZED batches: the value the reference's compute_mmd / compute_scoring_rule /
    compute_expected_scoring_rule (sigkernel.py:146-197) assemble from three (two) compute_Gram calls, from ONE Gram block
    K(X, Z), Z = [X; Y], and -- for the MMD -- the triangle of K(Y, Y).

    Why: a Gram matrix of a few thousand pairs is one wave's skew fill plus a pair or two per lane group, so the three forward and
    two adjoint launches of the reference's composition each leave most of the chip idle, and some sixty small launches (staging,
    reductions and their autograd) sit between them.  Here K_XX and K_XY are the column blocks of ONE forward launch (with the edges
    for backward) and their gradients ONE adjoint launch over the same pairs, weighted by the constant d loss / dK (_loss_weights;
    the reference's 2x rule for K_XX is in the weights); the reductions are two multiply-sums.  Same pairs, same kernels, same
    per-pair values; only the order in which the (A, A+B) values are summed differs from the reference's formula (last-bit).
    Without a gradient and with the K_YY term the whole triangle of K(Z, Z) is one launch."""

    @staticmethod
    def forward(ctx, X, Y, static_kernel, dyadic_order, _naive_solver, workspace_bytes, with_yy):
        be = _lib.get_backend()
        A, B = X.shape[0], Y.shape[0]
        Xd, Yd = X.detach().contiguous(), Y.detach().contiguous()
        ctx.static_kernel, ctx.dyadic_order, ctx._naive_solver, ctx.workspace_bytes = static_kernel, dyadic_order, _naive_solver, workspace_bytes
        ctx.kept_edges = ctx.K = ctx.launch = None
        need = ctx.needs_input_grad[0]
        fast = _loss_launch_ok(be, static_kernel, Xd, Yd, dyadic_order, _naive_solver, need, with_yy, workspace_bytes)
        if fast is not None:
            # the one-launch glue (csrc/sk_loss.hip): staging of [X; Y] (no concatenated copy), K(X, [X; Y]) and the strict triangle of
            # K(Y, Y) in ONE forward launch, the scalar in one reduction -- three launches where the route below issues a dozen
            res = be.loss_forward(fast[0], fast[1], Xd, Yd, dyadic_order, _naive_solver, with_yy, keep_edges=need)
            if res is not None:
                val, out, edges, staged, wb = res
                if need:
                    ctx.save_for_backward(X)
                    ctx.launch = (fast, out, edges, staged, wb, A, B, Xd.shape[1], Yd)
                return val
        Z = torch.cat((Xd, Yd))
        wf, wb, wy = _loss_weights(A, B, X.dtype, X.device)
        if not need and with_yy:
            K_ZZ = _gram_symmetric(be, static_kernel, Z, dyadic_order, _naive_solver, workspace_bytes)
            return (K_ZZ[:A] * wf).sum() + (K_ZZ[A:, A:] * wy).sum()
        # (K_YY on a side stream while a graph is captured -- compute_mmd's composition does that -- buys nothing here: replays of
        # 0.305 / 0.534 / 1.29 ms forked against 0.309 / 0.561 / 1.28 on one stream at 32 / 64 / 128 paths, tools/experiments/r04_merged_fork.py)
        K_YY = _gram_symmetric(be, static_kernel, Yd, dyadic_order, _naive_solver, workspace_bytes) if with_yy else None
        fused = _fused_static(static_kernel, True) is not None
        ctx.kept_edges = [] if need else None
        K_XZ = _gram_block(be, static_kernel, Xd, Z, dyadic_order, _naive_solver, workspace_bytes, (3 if fused else 8) if need else None,
                           ctx.kept_edges)
        val = (K_XZ * wf).sum()
        if with_yy:
            val = val + (K_YY * wy).sum()
        if need:
            ctx.save_for_backward(X, Z)
            ctx.K, ctx.wb = K_XZ, wb
        return val

    @staticmethod
    def backward(ctx, grad_output):
        be = _lib.get_backend()
        if ctx.launch is not None:
            # ONE fused adjoint over the rectangle's pairs, weighted by the constant d value / dK the forward's reduction left, from the
            # arrays the forward staged and the edges it kept; the upstream scalar stays on the device and multiplies the fold of the
            # partial sums into dL/dX (the adjoint is linear in it): screen + sweep + rescue + fold, four launches
            (X,) = ctx.saved_tensors
            (kind, param), out, edges, (Zr, Zt, Zr_adj), wb, A, B, M, Yd = ctx.launch
            Xd = X.detach().contiguous()
            gs = grad_output.detach().to(torch.float64).contiguous()
            adj = be.linear_adjoint_fused if kind == 0 else be.rbf_adjoint_fused
            res = adj(Xd, None, param, ctx.dyadic_order, edges, wb, gram=True, kfinal=out[:A * (A + B)], naive=ctx._naive_solver,
                      staged=(Zr_adj, Zt, A + B, M), gscale=gs)
            if res is None:
                # sk_route_query named the one-band adjoint for this shape and its launcher declined all the same (a scope or workspace
                # check the query does not mirror): like every other gradient route, fall back -- the rows' gradient from the paths
                # (its own forward sweep; the kept edges are of no use to the streaming kernels) with the same weights
                go = (wb.view(A, A + B) * gs).contiguous()
                g = _rows_gradient(be, ctx.static_kernel, Xd, torch.cat((Xd, Yd)), go, ctx.dyadic_order, ctx._naive_solver, True, None,
                                   ctx.workspace_bytes, None)
                return g, None, None, None, None, None, None
            return res[0], None, None, None, None, None, None
        X, Z = ctx.saved_tensors
        go = (ctx.wb * grad_output.to(X.dtype)).contiguous()
        kept, ctx.kept_edges = ctx.kept_edges, None
        grad_X = _rows_gradient(be, ctx.static_kernel, X.detach().contiguous(), Z, go, ctx.dyadic_order, ctx._naive_solver, True, kept,
                                ctx.workspace_bytes, ctx.K)
        ctx.K = None
        return grad_X, None, None, None, None, None, None


def _loss_launch_ok(be, static_kernel, Xd, Yd, dyadic, naive, need_grad, with_yy=True, workspace_bytes=None):
    """(kind, param) when a loss wrapper's call can take the one-launch glue of csrc/sk_loss.hip: exactly LinearKernel / RBFKernel,
    fp64 paths of one length, the ONE-BAND fused kernels for the forward and -- with a gradient pending -- for the adjoint too
    (sk_route_query on the rectangle's shape), pair counts inside the 32-bit / 15-bit fields of sk_solve_fwd_loss_f64 and
    sk_loss_value_f64 (checked HERE, before any launch), and -- with a gradient pending -- edges of the whole rectangle that fit the
    share of the transient budget a call may keep until backward (`keep_edges_fraction` of workspace_bytes, as _gram_block); None
    otherwise (the merged route's torch glue serves everything else: it tiles the rows and drops the edges where they do not fit)."""
    if routes.no_loss_launch or not hasattr(be, "loss_forward") or Xd.dtype != torch.float64 or Xd.shape[1:] != Yd.shape[1:]:
        return None
    fused = _fused_static(static_kernel, True)
    if fused is None:
        return None
    A, B, M = Xd.shape[0], Yd.shape[0], Xd.shape[1]
    tri_n = B if with_yy else 0
    p_rect, p_tri = A * (A + B), (tri_n * (tri_n - 1) // 2 if tri_n > 1 else 0)
    # sk_solve_fwd_loss_f64 packs A and tri_n into 15 bits each and indexes pairs with 31 bits; sk_loss_value_f64 indexes with 32
    if A > 0x7fff or tri_n > 0x7fff or p_rect >= 0x7ff00000 or p_rect + p_tri >= 0x7ff00000:
        return None
    if _route(be, OP_FORWARD, static_kernel, Xd, Yd, dyadic, naive, True) != FUSED:
        return None
    if need_grad and _route(be, OP_ADJOINT, static_kernel, Xd, Yd, dyadic, naive, True) != FUSED:
        return None
    # what the call allocates in one piece and (the edges, the weights, the staged paths) holds until backward: 8 (MM + NN + 32)
    # bytes of edges per rectangle pair, values + weights + pair table 8 bytes per pair each
    held = 8.0 * p_rect * ((2 * ((M - 1) << int(dyadic)) + 32) if need_grad else 0) + 8.0 * (3 * p_rect + 2 * p_tri)
    if (workspace_bytes is not None or held > _cost("loss_launch_free_bytes")) and \
            held > _cost("keep_edges_fraction") * _budget(Xd.device, workspace_bytes):
        return None
    return fused


_LOSS_LAUNCH_FREE_BYTES = None    # "loss_launch_free_bytes": below this the one-launch loss route does not ask the device for its free memory


class _NoGradCtx:
    """What the autograd Functions' forward needs of a context when no gradient can be asked for: the call skips
    torch.autograd.Function.apply (a quarter of the host time of a C1-sized call) and returns the same values."""
    needs_input_grad = (False,) * 8

    def save_for_backward(self, *tensors):
        pass


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


class SigKernel:
    """Signature kernel k_sig(x, y) = <S(f(x)), S(f(y))> for a static kernel k(x, y) = <f(x), f(y)>.

    Drop-in for the reference's ``SigKernel`` (sigkernel.py:15-197): same constructor, same methods,
    same shapes / dtypes / devices.  Tensors must live on a HIP device (``'cuda'`` in PyTorch-ROCm).

    Extra, optional: ``workspace_bytes`` bounds the transient HBM a call may use (default: half of
    the free memory, at most 48 GiB); ``process_group`` shards ``compute_Gram`` rows over the ranks
    of a ``torch.distributed`` group (see :mod:`sigkernel_amd.distributed`).
    """

    def __init__(self, static_kernel, dyadic_order, _naive_solver=False, workspace_bytes=None, process_group=None):
        self.static_kernel = static_kernel
        self.dyadic_order = dyadic_order
        self._naive_solver = _naive_solver
        self.workspace_bytes = workspace_bytes
        self.process_group = process_group

    def compute_kernel(self, X, Y, max_batch=100):
        """X (batch, len_x, dim), Y (batch, len_y, dim) -> (batch,) vector k(X^i_T, Y^i_T).

        ``max_batch`` is kept for signature compatibility (sigkernel.py:23); tiling is by HBM budget.  Under a process group the
        pairs are sharded over the ranks like Gram rows (sigkernel_amd.distributed.ShardedPaired)."""
        if self.process_group is not None:
            from .distributed import sharded_kernel
            return sharded_kernel(self, X, Y, self.process_group)
        if not _wants_grad(X, Y):
            return _SigKernel.forward(_NoGradCtx(), X, Y, self.static_kernel, self.dyadic_order, self._naive_solver, self.workspace_bytes)
        return _SigKernel.apply(X, Y, self.static_kernel, self.dyadic_order, self._naive_solver, self.workspace_bytes)

    def compute_kernel_and_derivatives_Gram(self, X, Y, gamma, max_batch=100):
        """X (batch_X, len_x, dim), Y (batch_Y, len_y, dim), gamma (batch_X, len_x, dim) -> three (batch_X, batch_Y)
        matrices: k(X^i, Y^j) and its first and second directional derivative along gamma^i (sigkernel.py:43-89)."""
        if self.process_group is not None:
            from .distributed import sharded_kgrad
            return sharded_kgrad(self, X, Y, gamma, self.process_group)
        return k_kgrad(X, Y, gamma, self.dyadic_order, self.static_kernel, workspace_bytes=self.workspace_bytes)

    def compute_Gram(self, X, Y, sym=False, max_batch=100):
        """X (batch_X, len_x, dim), Y (batch_Y, len_y, dim) -> (batch_X, batch_Y) matrix k(X^i_T, Y^j_T)."""
        if self.process_group is not None:
            from .distributed import sharded_gram
            return sharded_gram(self, X, Y, sym, self.process_group)
        if not _wants_grad(X, Y):
            return _SigKernelGram.forward(_NoGradCtx(), X, Y, self.static_kernel, self.dyadic_order, sym, self._naive_solver,
                                          self.workspace_bytes)
        return _SigKernelGram.apply(X, Y, self.static_kernel, self.dyadic_order, sym, self._naive_solver,
                                    self.workspace_bytes)

    def compute_distance(self, X, Y, max_batch=100):
        """(batch,) paired squared distances reduced to their mean, as the reference does (sigkernel.py:130-144)."""
        assert not Y.requires_grad, "the second input should not require grad"
        n = X.shape[0] if X.dim() == 3 else 0
        if (not routes.no_merged_loss and X.dim() == 3 and X.shape == Y.shape and X.dtype == Y.dtype and X.device == Y.device and n > 0
                and X.shape[1] >= 2 and float(n) * float((X.shape[1] - 1) << int(self.dyadic_order)) ** 2 < _cost("paired_merge_cells")):
            # training-sized batches: k(x_i, x_i) and k(x_i, y_i) as ONE paired batch of 2n pairs -- one forward and one adjoint launch
            # instead of two each (launch- and fill-bound at these sizes); the same per-pair values, and the gradient reaches X through
            # the first argument of both halves, as in the reference's three calls (_SigKernel returns none for a second argument)
            K2 = self.compute_kernel(torch.cat((X, X)), torch.cat((X.detach(), Y)), max_batch)
            K_YY = self.compute_kernel(Y, Y, max_batch)
            return torch.mean(K2[:n]) + torch.mean(K_YY) - 2. * torch.mean(K2[n:])
        K_XX = self.compute_kernel(X, X, max_batch)
        K_YY = self.compute_kernel(Y, Y, max_batch)
        K_XY = self.compute_kernel(X, Y, max_batch)
        return torch.mean(K_XX) + torch.mean(K_YY) - 2. * torch.mean(K_XY)

    def compute_scoring_rule(self, X, y, max_batch=100):
        """S(X, y) = E[k(X, X)] - 2 E[k(X, y)] with y of shape (1, len_y, dim) (sigkernel.py:146-161)."""
        assert not y.requires_grad, "the second input should not require grad"
        merged = self._merged_loss(X, y, with_yy=False)
        if merged is not None:
            return merged
        K_XX = self.compute_Gram(X, X, sym=True, max_batch=max_batch)
        K_Xy = self.compute_Gram(X, y, sym=False, max_batch=max_batch)
        K_XX_m = (torch.sum(K_XX) - torch.sum(torch.diag(K_XX))) / (K_XX.shape[0] * (K_XX.shape[0] - 1.))
        return K_XX_m - 2. * torch.mean(K_Xy)

    def compute_expected_scoring_rule(self, X, Y, max_batch=100):
        """S(X, Y) = E_Y[S(X, y)] (sigkernel.py:163-178)."""
        assert not Y.requires_grad, "the second input should not require grad"
        merged = self._merged_loss(X, Y, with_yy=False)
        if merged is not None:
            return merged
        K_XX = self.compute_Gram(X, X, sym=True, max_batch=max_batch)
        K_XY = self.compute_Gram(X, Y, sym=False, max_batch=max_batch)
        K_XX_m = (torch.sum(K_XX) - torch.sum(torch.diag(K_XX))) / (K_XX.shape[0] * (K_XX.shape[0] - 1.))
        return K_XX_m - 2. * torch.mean(K_XY)

    def _merged_loss(self, X, Y, with_yy):
        """The loss wrappers' merged route for training-sized batches (_SigKernelLoss: one forward and one adjoint launch over
        K(X, [X; Y])), or None where it does not apply -- paths of different lengths, a process group, batches whose K_XX the
        composition solves as a blocked triangle (_SYM_MIN_CELLS grid cells and more: that saves more than the merged launches
        do; below it, measured on one box from 16 to 512 paths, tools/experiments/r04_merged_loss.py: merged 0.30 / 0.31 / 0.56 /
        1.27 / 4.16 / 15.1 ms against 0.58 / 0.58 / 0.80 / 1.57 / 4.48 / 15.5 at 16 / 32 / 64 / 128 / 256 / 512 paths of
        BASELINE configs[1]'s shape), fewer than two paths (the reference's 0 / 0), malformed inputs (the Gram calls raise for
        those), `routes.no_merged_loss`."""
        if routes.no_merged_loss or self.process_group is not None or X.dim() != 3 or Y.dim() != 3:
            return None
        A, B = X.shape[0], Y.shape[0]
        if X.shape[1:] != Y.shape[1:] or X.shape[1] < 2 or A < 2 or B < (2 if with_yy else 1) or X.dtype != Y.dtype or X.device != Y.device:
            return None
        if float(A) * A * float((X.shape[1] - 1) << int(self.dyadic_order)) ** 2 >= _cost("sym_min_cells"):
            return None
        args = (X, Y, self.static_kernel, self.dyadic_order, self._naive_solver, self.workspace_bytes, with_yy)
        if not _wants_grad(X):
            return _SigKernelLoss.forward(_NoGradCtx(), *args)
        return _SigKernelLoss.apply(*args)

    def compute_mmd(self, X, Y, max_batch=100):
        """Unbiased MMD^2 between the samples X and Y (sigkernel.py:180-197).

        Training-sized batches (a few thousand pairs per Gram matrix) leave most of the chip idle: each of the three forward
        launches -- and, in backward, of the two adjoint launches -- is a wave's skew fill plus a pair or two.  While a hipGraph is
        being CAPTURED (`torch.cuda.graph`: what a training loop with static shapes should replay) the three Gram matrices are
        therefore put on three streams -- K_YY and K_XY fork from the caller's stream and join it before the reductions; autograd
        runs each matrix's backward on the stream of its forward -- so the graph holds them as parallel branches: replays 10-15 %
        faster at 16..64 paths (tools/experiments/r04_mmd_streams.py), bit-identical gradients.  Eager calls stay on one stream:
        they are bound by the host's launch rate there, and the stream switches cost more than the overlap returns (0.59 -> 0.75 ms
        at 32 paths); large batches fill the chip with one launch (round 3: no gain from streams at BASELINE configs[3])."""
        assert not Y.requires_grad, "the second input should not require grad"
        merged = self._merged_loss(X, Y, with_yy=True)
        if merged is not None:
            return merged
        small = (X.is_cuda and Y.is_cuda and self.process_group is None and routes_allow_streams()
                 and max(X.shape[0], Y.shape[0]) ** 2 <= _cost("mmd_streams_max_pairs") and X.shape[0] > 1 and Y.shape[0] > 1)
        if small:
            s_yy, s_xy = _side_streams(X.device)      # (created by the warm-up calls that precede a capture, never inside one)
        if small and torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream(X.device)
            s_yy.wait_stream(cur)
            s_xy.wait_stream(cur)
            with torch.cuda.stream(s_yy):
                K_YY = self.compute_Gram(Y, Y, sym=True, max_batch=max_batch)
            with torch.cuda.stream(s_xy):
                K_XY = self.compute_Gram(X, Y, sym=False, max_batch=max_batch)
            K_XX = self.compute_Gram(X, X, sym=True, max_batch=max_batch)
            cur.wait_stream(s_yy)
            cur.wait_stream(s_xy)
            K_YY.record_stream(cur)       # (allocated on the side streams, consumed on the caller's)
            K_XY.record_stream(cur)
        else:
            K_XX = self.compute_Gram(X, X, sym=True, max_batch=max_batch)
            K_YY = self.compute_Gram(Y, Y, sym=True, max_batch=max_batch)
            K_XY = self.compute_Gram(X, Y, sym=False, max_batch=max_batch)
        K_XX_m = (torch.sum(K_XX) - torch.sum(torch.diag(K_XX))) / (K_XX.shape[0] * (K_XX.shape[0] - 1.))
        K_YY_m = (torch.sum(K_YY) - torch.sum(torch.diag(K_YY))) / (K_YY.shape[0] * (K_YY.shape[0] - 1.))
        return K_XX_m + K_YY_m - 2. * torch.mean(K_XY)
