"""ctypes binding of libsigkernel_amd.so (C ABI: include/sigkernel_amd.h).

There is no CPU fallback: if the shared library is missing, or a tensor is not on a HIP
device, the calls raise.  torch is used for device memory and streams only.
"""
import ctypes
import os

import torch

from ._routes import routes

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsigkernel_amd.so")

SK_OK = 0
# sk_route_query: operations and answers (include/sigkernel_amd.h)
OP_FORWARD, OP_ADJOINT, OP_ADJOINT_SYM = 0, 1, 2
ROUTE_STREAM, ROUTE_FUSED, ROUTE_FUSED_MB, ROUTE_FUSED_MB_SWAP, ROUTE_FUSED_SWAP = 0, 1, 2, 3, 4
ROUTE_NO_STREAM = 1
ROUTE_NO_SWAP = 2
SCHEME_DEFAULT = 0
SCHEME_NAIVE = 1
FLAG_EXACT = 1
FLAG_SIMPLE = 2
FLAG_FAST_ONLY = 4
FLAG_EDGES_GIVEN = 8

_lib = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/sigkernel_amd.h one to one
SIGNATURES = {
    "sk_version": (_int, []),
    "sk_build_info": (ctypes.c_char_p, []),
    "sk_route_query": (_int, [_int, _int, _int, _int, _int, _int, _int, _int, _int]),
    "sk_cost_query": (ctypes.c_double, [_int]),
    "sk_cost_name": (ctypes.c_char_p, [_int]),
    "sk_cost_note": (ctypes.c_char_p, [_int]),
    "sk_solve_fwd_static_cols": (_int, [_int, _int]),
    "sk_reload_knobs": (None, []),
    "sk_launch_trace": (_int, [_int]),
    "sk_launch_trace_dump": (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t, _int]),
    "sk_linear_prescale": (ctypes.c_double, [_int]),
    "sk_status_string": (ctypes.c_char_p, [_int]),
    "sk_device_count": (_int, []),
    "sk_increments_f64": (_int, [_vp, _i64, _int, _int, _vp, _i64, _vp]),
    "sk_increments_f32": (_int, [_vp, _i64, _int, _int, _vp, _i64, _vp]),
    "sk_static_increments_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _i64, _i64, _int, _int, _int, _vp, _i64, _vp]),
    "sk_static_increments_f32": (_int, [_int, ctypes.c_double, _vp, _vp, _i64, _i64, _int, _int, _int, _vp, _i64, _vp]),
    "sk_static_adjoint_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "sk_static_adjoint_f32": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "sk_linear_adjoint_f64": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "sk_linear_adjoint_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "sk_static_adjoint2_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _int, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _int,
                                      _vp, _vp]),
    "sk_static_adjoint2_f32": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _int, _vp, _i64, _vp, _i64, _i64, _int, _int, _int, _int,
                                      _vp, _vp]),
    "sk_increments_adjoint_f64": (_int, [_vp, _i64, _vp, _i64, _int, _int, _vp, _vp]),
    "sk_increments_adjoint_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _vp, _vp]),
    "sk_solve_fwd_f64": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "sk_solve_fwd_f32": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "sk_solve_fwd_linear_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_fwd_linear_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_fwd_rbf_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_solve_fwd_rbf_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_prep_pair_f64": (_int, [_vp, _i64, _int, _vp, _i64, _int, _int, _int, ctypes.c_double, ctypes.c_double, _vp, _int, _vp, _int, _int, _vp]),
    "sk_prep_pair_f32": (_int, [_vp, _i64, _int, _vp, _i64, _int, _int, _int, ctypes.c_double, ctypes.c_double, _vp, _int, _vp, _int, _int, _vp]),
    "sk_solve_fwd_static_workspace_bytes": (_sz, [_int, _i64, _int, _int, _int, _int]),
    "sk_solve_fwd_static_rows": (_int, [_int, _int, _int]),
    "sk_solve_fwd_static_split": (_int, [_int, _i64, _int, _int, _int, _int]),
    "sk_solve_fwd_static_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp,
                                       _vp, _vp, _sz, _vp]),
    "sk_solve_fwd_static_f32": (_int, [_int, ctypes.c_double, _vp, _vp, _int, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int,
                                       _vp, _vp, _vp, _sz, _vp]),
    "sk_solve_deriv_static_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int, _vp]),
    "sk_solve_deriv_static_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int,
                                         ctypes.c_double, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sk_linear_adjoint_fused_mb_layout": (_int, [_i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "sk_linear_adjoint_fused_mb_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _sz, _vp, _vp,
                                              _sz, _vp, ctypes.c_double, ctypes.c_double, _vp, _sz, _vp]),
    "sk_rbf_adjoint_fused_mb_layout": (_int, [_i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sk_rbf_adjoint_fused_mb_f64": (_int, [_vp, _vp, _int, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp,
                                           _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _sz, _vp]),
    "sk_solve_fwd_linear_sym_f64": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_fwd_linear_sym_f32": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_fwd_rbf_sym_f64": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_solve_fwd_rbf_sym_f32": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_solve_fwd_rbf_edges_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "sk_solve_fwd_linear_edges_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "sk_linear_adjoint_fused_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp,
                                           _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _sz, _vp]),
    "sk_rbf_adjoint_fused_f64": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp,
                                        _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _sz, _vp]),
    "sk_fused_rescue_workspace_bytes": (_sz, [_int, _i64, _int, _int, _int, _int]),
    "sk_adj_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int, _int]),
    "sk_adj_rescue_slot_bytes": (_sz, [_int, _int, _int]),
    "sk_plan_wave_shares": (_int, [_i64, _int, _i64, _i64, _int, _int, _vp, _vp, _vp]),
    "sk_plan_group_chunks": (_int, [_i64, _i64, _i64, _i64, _int, _int, _int, _i64, _vp, _vp, _vp]),
    "sk_adj_rescue_f64": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _vp, ctypes.c_double, _vp, _vp, _i64, _vp, _sz, _vp]),
    "sk_adj_rescue_f32": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _vp, ctypes.c_double, _vp, _vp, _i64, _vp, _sz, _vp]),
    "sk_prep_paths_f64": (_int, [_vp, _i64, _int, _int, _int, _int, ctypes.c_double, _vp, _int, _int, _vp]),
    "sk_prep_paths_f32": (_int, [_vp, _i64, _int, _int, _int, _int, ctypes.c_double, _vp, _int, _int, _vp]),
    "sk_strip_edges_bytes": (_sz, [_i64, _int, _int, _int, _int]),
    "sk_solve_fwd_edges_f64": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_fwd_edges_f32": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp]),
    "sk_solve_adj_f64": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sk_solve_adj_f32": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sk_deriv_increments_f64": (_int, [_vp, _vp, _vp, ctypes.c_double, _i64, _int, _int, _vp, _vp, _vp, _i64, _vp]),
    "sk_deriv_increments_f32": (_int, [_vp, _vp, _vp, ctypes.c_double, _i64, _int, _int, _vp, _vp, _vp, _i64, _vp]),
    "sk_static_deriv_increments_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int,
                                              ctypes.c_double, _vp, _vp, _vp, _i64, _vp]),
    "sk_static_deriv_increments_f32": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int,
                                              ctypes.c_double, _vp, _vp, _vp, _i64, _vp]),
    "sk_solve_deriv_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "sk_prep_cat_f64": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, ctypes.c_double, ctypes.c_double, _vp, _vp, _int, _vp, _int, _int, _vp, _i64, _vp]),
    "sk_prep_cat_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, ctypes.c_double, ctypes.c_double, _vp, _vp, _int, _vp, _int, _int, _vp, _i64, _vp]),
    "sk_solve_fwd_loss_f64": (_int, [_int, ctypes.c_double, _vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "sk_loss_value_f64": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "sk_loss_weights_f64": (_int, [_i64, _i64, _vp, _vp, _vp]),
    "sk_rbf_adjoint_finish_f64": (_int, [_vp, _i64, _i64, _int, _int, _vp, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_linear_adjoint_finish_f64": (_int, [_vp, _i64, _i64, _int, _int, _int, ctypes.c_double, _vp, _vp, _vp]),
    "sk_solve_deriv_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
}


class SigKernelLibraryError(RuntimeError):
    pass


def _stage_rows(kind, M, gram):
    """Rows per staged first path of the one-band kernels ([A][rows][8] fp64).  A Gram call stages A + B paths -- nothing -- and keeps the
    256 every variant accepts; a PAIRED batch stages every pair's x, and the padding was most of its HBM traffic (262 144 pairs of 64
    points: 16 KB per path, three staging launches of 1.6 ms in a 18 ms training step): the smallest of 64 / 128 / 256 rows that holds
    lanes x rows per lane of every variant (linear: M - 1 increments; rbf: M nodes and the node row above)."""
    if gram:
        return 256
    need = (M - 1) if kind == 0 else 2 * M
    return 64 if need <= 64 else (128 if need <= 128 else 256)


ABI_VERSION = 330      # include/sigkernel_amd.h: sk_version()


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SigKernelLibraryError(
                "%s not found: build it with `python -m sigkernel_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.sk_version() != ABI_VERSION:      # (a stale in-tree build: the signatures above would bind the wrong arguments)
            raise SigKernelLibraryError("%s exports ABI %d, this binding is written against %d: rebuild it (python -m sigkernel_amd.build)"
                                        % (LIB_PATH, lib.sk_version(), ABI_VERSION))
        _lib = lib
    return _lib


_COSTS = None


def costs():
    """{name: (value, note)} -- the library's cost table (sk_cost_query, csrc/sk_route.hip): every measured crossover the launchers and
    the host layer decide by, with the measurement each came from."""
    global _COSTS
    if _COSTS is None:
        lib, table, i = load(), {}, 0
        while True:
            name = lib.sk_cost_name(i)
            if not name:
                break
            table[name.decode()] = (float(lib.sk_cost_query(i)), lib.sk_cost_note(i).decode())
            i += 1
        _COSTS = table
    return _COSTS


def cost(name):
    return costs()[name][0]


def launch_trace(enable=True):
    """Count every kernel launch of the library per kernel instance from now on (sk_launch_trace); returns the previous state."""
    return bool(load().sk_launch_trace(1 if enable else 0))


def launch_counts(reset=False):
    """{device symbol (mangled): launches} since the last reset (sk_launch_trace_dump)."""
    lib = load()
    n = int(lib.sk_launch_trace_dump(None, 0, 0))
    buf = ctypes.create_string_buffer(n + 16)
    lib.sk_launch_trace_dump(buf, n + 16, 1 if reset else 0)
    out = {}
    for ln in buf.value.decode().split("\n"):
        if "\t" in ln:
            c, name = ln.split("\t", 1)
            out[name] = out.get(name, 0) + int(c)
    return out


class _SplitStatus:
    """The status word of a split-mode sk_solve_fwd_static_* launch, reduced on demand (`int(s)` synchronises): how many (band, pair)
    items gave up waiting for the band above (their pairs are NaN); 0 after every normal launch."""

    def __init__(self, ws):
        self._ws = ws

    def __int__(self):
        n8 = self._ws.numel() & ~7
        return int(self._ws[n8 - 8:n8].view(torch.int64).item())


def _check(status, what):
    if status != SK_OK:
        msg = load().sk_status_string(status).decode()
        exc = ValueError if status in (1, 2) else RuntimeError
        raise exc("%s: %s (sk_status %d)" % (what, msg, status))


def _suffix(t):
    if t.dtype == torch.float64:
        return "f64"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError("sigkernel_amd supports float64 and float32 tensors, got %s" % t.dtype)


def _dev(t, name):
    if t.device.type != "cuda":
        raise RuntimeError(
            "sigkernel_amd runs on MI355X only: tensor `%s` is on %s; move it to a HIP device "
            "('cuda' in PyTorch-ROCm). There is no CPU fallback." % (name, t.device))
    if not t.is_contiguous():
        raise ValueError("tensor `%s` must be contiguous" % name)
    return t


def _rbf_extra_row(Mc, dyadic):
    """1 when the one-band RBF kernels keep their edges in the layout of a strip one row taller (csrc/sk_internal.h: rbf_edge_geom)."""
    rc = 4 >> min(int(dyadic), 2)
    cap = 8 * rc
    while cap < Mc:
        cap *= 2
    return 1 if (cap == Mc and cap < 64 * rc) else 0


def _padded_ld(n, elem_size):
    """Row stride (elements) that pads every increment row to whole 128-byte cache lines: the LDS-DMA kernels
    fetch rows line by line, and the adjoint sweep needs the (zero) padding to exist."""
    q = 128 // elem_size
    return (n + q - 1) // q * q


def _row_stride(t, name):
    """inc_c [..., Mc, Nc] with unit column stride and batch dims packed over rows of stride ld >= Nc.
    Returns (tensor, ld); anything else is densified."""
    if t.device.type != "cuda":
        _dev(t, name)
    Mc, Nc = t.shape[-2:]
    if t.dim() >= 2 and t.stride(-1) == 1 and t.stride(-2) >= Nc:
        ld = t.stride(-2)
        expect, ok = Mc * ld, True
        for size, stride in zip(reversed(t.shape[:-2]), reversed(t.stride()[:-2])):
            if size != 1 and stride != expect:
                ok = False
                break
            expect *= size
        if ok:
            return t, ld
    t = t.contiguous()
    return t, Nc


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The raw handle of torch's current stream on t's device (the fast C entry point where torch has it: a C1-sized call spent a
    third of its host time constructing Stream objects)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device(dev):
    """`with _device(dev):` -- torch.cuda.device(dev), or nothing at all when dev is the current device already (the usual case: the
    guard's two device switches are a measurable part of a small call)."""
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _queue(dev, pairs=1 << 30):
    """64 bytes of device scratch for one launch's work counter (the library zeroes it on the launch stream); the caching
    allocator keeps it alive for the work queued on this stream and hands it out again afterwards.  None for launches too small to
    fill the chip (the library then deals out equal static shares, as it would anyway)."""
    return torch.empty(8, dtype=torch.int64, device=dev) if pairs >= 16384 else None


def _prep_paths(X, diff, dim_major, scale, rows, fd=8):
    """fp64, zero-padded staging of a path batch for the fused kernels in ONE launch (sk_prep_paths_*):
    X (A,M,D) -> [A][rows][fd] (dim_major=False) or [A][fd][rows] (dim_major=True) of scale*(x[p+1]-x[p]) (diff) or
    scale*x[p]."""
    A, M, D = X.shape
    out = torch.empty((A, fd, rows) if dim_major else (A, rows, fd), dtype=torch.float64, device=X.device)
    fn = getattr(load(), "sk_prep_paths_" + _suffix(X))
    _check(fn(_ptr(X), A, M, D, int(bool(diff)), int(bool(dim_major)), float(scale), _ptr(out), int(rows), int(fd), _stream(X)),
           "sk_prep_paths")
    return out


def _prep_pair(X, Y, diff, scale_x, rows_x, rows_y, fd=8):
    """Both staged arrays of a call from ONE allocation and ONE launch (sk_prep_pair_*): ([A][rows_x][fd] from X scaled by scale_x,
    [B][fd][rows_y] from Y)."""
    A, M, D = X.shape
    B, N = Y.shape[0], Y.shape[1]
    nx = A * rows_x * fd
    buf = torch.empty(nx + B * fd * rows_y, dtype=torch.float64, device=X.device)
    out_x, out_y = buf[:nx].view(A, rows_x, fd), buf[nx:].view(B, fd, rows_y)
    fn = getattr(load(), "sk_prep_pair_" + _suffix(X))
    _check(fn(_ptr(X), A, M, _ptr(Y), B, N, D, int(bool(diff)), float(scale_x), 1.0, _ptr(out_x), int(rows_x), _ptr(out_y), int(rows_y), int(fd),
              _stream(X)), "sk_prep_pair")
    return out_x, out_y


class _WorstResidual:
    """The worst self-check residual of a fused-adjoint launch (NaN-propagating, as torch.max is), reduced only when somebody asks
    for it -- `float(r)` or `r.tensor()`: it is a diagnostic, and a backward pass should not pay a reduction kernel for it.  The
    per-pair residuals themselves are in `HipBackend.last_fused_err` (entries of -1: pairs the rescue took out of the sweep)."""
    __slots__ = ("err",)

    def __init__(self, err):
        self.err = err

    def tensor(self):
        return self.err.max()

    def __float__(self):
        return float(self.err.max())


class HipBackend:
    """The product back-end: every method enqueues HIP kernels on the current stream."""

    name = "hip"
    last_split_status = None      # _SplitStatus of the last split-mode sk_solve_fwd_static_* launch (None: the last one was not)

    @staticmethod
    def route(op, kind, D, M, N, dyadic, naive, elem_size, no_stream=False, no_swap=False):
        """Which kernel family serves the call (sk_route_query, csrc/sk_route.hip): ROUTE_STREAM / _FUSED / _FUSED_MB / _FUSED_MB_SWAP / _FUSED_SWAP.
        no_stream: never STREAM where a fused kernel exists (by default short paths, on which the multi-band kernels would mostly
        sweep padding, take the faster streaming route)."""
        return int(load().sk_route_query(int(op), int(kind), int(D), int(M), int(N), int(dyadic), SCHEME_NAIVE if naive else SCHEME_DEFAULT,
                                         int(elem_size), (ROUTE_NO_STREAM if no_stream else 0) | (ROUTE_NO_SWAP if no_swap else 0)))

    def increments(self, G):
        """G [..., M, N] -> inc_c [..., M-1, N-1] (sigkernel.py:217, :363)."""
        _dev(G, "G")
        M, N = G.shape[-2:]
        P = G.numel() // (M * N)
        ld = _padded_ld(N - 1, G.element_size())
        out = torch.empty(G.shape[:-2] + (M - 1, ld), dtype=G.dtype, device=G.device)
        with _device(G.device):
            fn = getattr(load(), "sk_increments_" + _suffix(G))
            _check(fn(_ptr(G), P, M, N, _ptr(out), ld, _stream(G)), "sk_increments")
        return out[..., : N - 1]      # rows stay 16-byte aligned underneath (stride(-2) == ld)

    MAX_FUSED_DIM = 32

    def static_increments(self, kind, param, X, Y, gram):
        """Static kernel + increments in one pass (kind 0 = linear with param = scale, 1 = rbf with param = sigma).

        gram=True: X (A,M,D), Y (B,N,D) -> inc_c (A,B,M-1,N-1); gram=False (paired): Y (A,N,D) -> (A,M-1,N-1).
        Returns None when the shape is outside the fused kernels (D > 32): the caller takes the generic path."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        if D > self.MAX_FUSED_DIM or M < 2 or N < 2:
            return None
        ld = _padded_ld(N - 1, X.element_size())
        shape = (A, B, M - 1, ld) if gram else (A, M - 1, ld)
        out = torch.empty(shape, dtype=X.dtype, device=X.device)
        with _device(X.device):
            fn = getattr(load(), "sk_static_increments_" + _suffix(X))
            _check(fn(int(kind), float(param), _ptr(X), _ptr(Y), A, B if gram else 0, M, N, D, _ptr(out), ld, _stream(X)),
                   "sk_static_increments")
        return out[..., : N - 1]

    def solve_fwd_fused_linear(self, X, Y, scale, dyadic, naive, gram, keep_edges=False):
        """K[MM][NN] for the LINEAR static kernel with the increments formed inside the solver (nothing of size
        pairs x M x N in HBM).  Returns None outside the kernel's scope (dim > 8, dyadic > 2, more than one band).
        keep_edges: returns (K, edges) with the terminal row/column of every pair for solve_adj(..., edges=...)
        or the fused adjoint (edges None where that is not available: fp32 tensors -- the adjoint sweeps up-cast paths and forms
        its own edges)."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 8 or dyadic > 2 or Mc < 1 or Nc < 1 or Mc > 64 * (4 >> min(dyadic, 2)):
            return None
        Mrows, Ncp = _stage_rows(0, M, gram), (Nc + 15) // 16 * 16
        dev = X.device
        out = torch.empty((A, B) if gram else (A,), dtype=X.dtype, device=dev)
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        lib = load()
        with _device(dev):
            kappa = float(lib.sk_linear_prescale(int(dyadic)))             # 4^-d / sqrt(12): see sk_solve_fwd_linear_*
            # kappa s^2 (x[p+1] - x[p]), [A][256][8];  y[q+1] - y[q], dimension-major [B][8][Ncp]
            dXr, dYt = _prep_pair(X, Y, True, kappa * float(scale) ** 2, Mrows, Ncp)
            if keep_edges and X.dtype == torch.float64 and 0 <= dyadic <= 2:
                P = A * B if gram else A
                nbytes = int(lib.sk_strip_edges_bytes(P, Mc, Nc, int(dyadic), 8))
                if nbytes:
                    edges = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
                    rc = lib.sk_solve_fwd_linear_edges_f64(_ptr(dXr), _ptr(dYt), A, B if gram else 0, Mrows, Mc, Nc, Ncp, D,
                                                           int(dyadic), scheme, _ptr(out), _ptr(edges), _ptr(_queue(dev, A * B if gram else A)), _stream(X))
                    if rc == SK_OK:
                        return out, edges
                    if rc != 2:
                        _check(rc, "sk_solve_fwd_linear_edges")
            fn = getattr(lib, "sk_solve_fwd_linear_" + _suffix(X))
            rc = fn(_ptr(dXr), _ptr(dYt), A, B if gram else 0, Mrows, Mc, Nc, Ncp, D, int(dyadic), scheme, _ptr(out), _ptr(_queue(dev, A * B if gram else A)), _stream(X))
        if rc == 2:
            return None
        _check(rc, "sk_solve_fwd_linear")
        return (out, None) if keep_edges else out

    def solve_fwd_fused_rbf(self, X, Y, sigma, dyadic, naive, gram, keep_edges=False):
        """K[MM][NN] for the RBF static kernel with nodes and increments formed inside the solver (nothing of size
        pairs x M x N in HBM).  Returns None outside the kernel's scope (dim > 8, dyadic > 2, more than one band).
        keep_edges: (K, edges) as solve_fwd_fused_linear."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 8 or dyadic > 2 or Mc < 1 or Nc < 1 or M > 64 * (4 >> min(dyadic, 2)) or not float(sigma) > 0:
            return None
        Mrows, Ncp = _stage_rows(1, M, gram), (N + 15) // 16 * 16
        dev = X.device
        out = torch.empty((A, B) if gram else (A,), dtype=X.dtype, device=dev)
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        lib = load()
        with _device(dev):
            Xr, Yt = _prep_pair(X, Y, False, 1.0, Mrows, Ncp)     # the path points, [A][256][8]; dimension-major [B][8][Ncp]
            if keep_edges and X.dtype == torch.float64:
                P = A * B if gram else A
                # (node columns: when N - 1 is a multiple of 16 the strip is one column -- one line of units -- wider: rbf_edge_geom)
                # (node rows: when M - 1 fills the strip's lanes exactly -- RC 2^k rows -- the strip is one row taller: rbf_edge_geom)
                nbytes = int(lib.sk_strip_edges_bytes(P, Mc + _rbf_extra_row(Mc, dyadic), Nc + (1 if Nc % 16 == 0 else 0), int(dyadic), 8))
                if nbytes:
                    edges = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
                    rc = lib.sk_solve_fwd_rbf_edges_f64(_ptr(Xr), _ptr(Yt), A, B if gram else 0, Mrows, Mc, Nc, Ncp, D, int(dyadic),
                                                        scheme, 1.0 / float(sigma), _ptr(out), _ptr(edges), _ptr(_queue(dev, A * B if gram else A)), _stream(X))
                    if rc == SK_OK:
                        return out, edges
                    if rc != 2:
                        _check(rc, "sk_solve_fwd_rbf_edges")
            fn = getattr(lib, "sk_solve_fwd_rbf_" + _suffix(X))
            rc = fn(_ptr(Xr), _ptr(Yt), A, B if gram else 0, Mrows, Mc, Nc, Ncp, D, int(dyadic), scheme, 1.0 / float(sigma),
                    _ptr(out), _ptr(_queue(dev, A * B if gram else A)), _stream(X))
        if rc == 2:
            return None
        _check(rc, "sk_solve_fwd_rbf")
        return (out, None) if keep_edges else out

    def loss_forward(self, kind, param, X, Y, dyadic, naive, with_yy, keep_edges):
        """The loss wrappers' forward in THREE launches (csrc/sk_loss.hip): [X; Y] staged in both layouts straight from the two
        batches (sk_prep_cat_*), the rectangle K(X, [X; Y]) and -- with_yy -- the strict triangle of K(Y, Y) in ONE fused forward
        launch (sk_solve_fwd_loss_f64), the scalar K_XX_m - 2 mean(K_XY) [+ K_YY_m] (sk_loss_value_f64).  fp64 paths of one length
        within the one-band fused kernels' scope; None otherwise.  Returns (value 0-dim, out [P] in pair order, edges of the
        rectangle pairs or None, staged = (rows of the forward, cols, rows of the adjoint), wb = d value / dK of the rectangle with the
        reference's 2x rule on the K_XX block, or None without edges)."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B = Y.shape[0]
        Mc = M - 1
        rows_needed = Mc if kind == 0 else M
        if X.dtype != torch.float64 or Y.dtype != torch.float64 or Y.shape[1] != M or D > 8 or dyadic > 2 or Mc < 1 or A < 1 or B < 1:
            return None
        if rows_needed > 64 * (4 >> min(dyadic, 2)) or (kind == 1 and not float(param) > 0):
            return None
        dev = X.device
        lib = load()
        Mrows, Ncp = 256, ((Mc if kind == 0 else M) + 15) // 16 * 16
        Z = A + B
        P_rect = A * Z
        P = P_rect + (B * (B - 1) // 2 if with_yy else 0)
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        with _device(dev):
            two_rows = kind == 0 and keep_edges      # the linear adjoint's rows carry s^2, the forward's kappa s^2
            nr, nt = Z * Mrows * 8, Z * 8 * Ncp
            # one allocation: rows [| rows of the adjoint] | columns | the pair table [P][2] int32 right behind the columns
            buf = torch.empty(nr * (2 if two_rows else 1) + nt + P, dtype=torch.float64, device=dev)
            Zr = buf[:nr].view(Z, Mrows, 8)
            Zr2 = buf[nr:2 * nr].view(Z, Mrows, 8) if two_rows else None
            t0 = nr * (2 if two_rows else 1)
            Zt = buf[t0:t0 + nt].view(Z, 8, Ncp)
            tab = buf[t0 + nt:]
            if kind == 0:
                kappa = float(lib.sk_linear_prescale(int(dyadic)))
                s_rows, s_rows2, diff = kappa * float(param) ** 2, float(param) ** 2, 1
            else:
                s_rows, s_rows2, diff = 1.0, 1.0, 0
            _check(lib.sk_prep_cat_f64(_ptr(X), A, _ptr(Y), B, M, D, diff, s_rows, s_rows2, _ptr(Zr), _ptr(Zr2), Mrows, _ptr(Zt), Ncp, 8,
                                       _ptr(tab), B if with_yy else 0, _stream(X)), "sk_prep_cat")
            edges = None
            if keep_edges:
                if kind == 0:
                    nbytes = int(lib.sk_strip_edges_bytes(P_rect, Mc, Mc, int(dyadic), 8))
                else:
                    nbytes = int(lib.sk_strip_edges_bytes(P_rect, Mc + _rbf_extra_row(Mc, dyadic), Mc + (1 if Mc % 16 == 0 else 0), int(dyadic), 8))
                if not nbytes:
                    return None
                edges = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
            out = torch.empty(P, dtype=torch.float64, device=dev)
            rc = lib.sk_solve_fwd_loss_f64(int(kind), 1.0 / float(param) if kind == 1 else 0.0, _ptr(Zr), _ptr(Zt), _ptr(tab), A, B, B if with_yy else 0,
                                           Mrows, Mc, Mc, Ncp, D, int(dyadic), scheme, _ptr(out), _ptr(edges), _ptr(_queue(dev, P)), _stream(X))
            if rc == 2:
                return None
            _check(rc, "sk_solve_fwd_loss")
            value = torch.empty((), dtype=torch.float64, device=dev)
            wb = torch.empty(P_rect, dtype=torch.float64, device=dev) if keep_edges else None   # d value / dK: the adjoint's weights
            rc = lib.sk_loss_value_f64(_ptr(out), A, B, int(bool(with_yy)), _ptr(value), _ptr(wb), _stream(X))
            if rc == 2:          # (a decline, like the forward's: the caller's torch glue serves the call)
                return None
            _check(rc, "sk_loss_value")
        return value, out, edges, (Zr, Zt, Zr2 if two_rows else Zr), wb

    @staticmethod
    def loss_weights(A, B, grad_output, dev):
        """grad_output (a 0-dim device tensor) times d value / dK of the rectangle K(X, [X; Y]), the K_XX block doubled
        (sk_loss_weights_f64): the upstream gradient of the fused adjoint, one launch, nothing read back."""
        go = torch.empty(A * (A + B), dtype=torch.float64, device=dev)
        g = grad_output.detach().to(torch.float64).contiguous()
        with _device(dev):
            _check(load().sk_loss_weights_f64(A, B, _ptr(g), _ptr(go), _stream(go)), "sk_loss_weights")
        return go

    SYM_TABLE_MAX_PAIRS = 1 << 26

    def solve_fwd_fused_sym(self, kind, param, X, dyadic, naive):
        """Symmetric Gram matrix K[a][b] = k(x_a, x_b) of ONE batch with the static kernel formed inside the solver: only the pairs
        on and above the diagonal are solved, in one launch, each written twice (sk_solve_fwd_linear_sym_* / sk_solve_fwd_rbf_sym_*).
        None outside the single-band fused kernels' scope (dim > 8, dyadic > 2, more than one band)."""
        _dev(X, "X")
        A, M, D = X.shape
        Mc = M - 1
        rows = Mc if kind == 0 else M
        if D > 8 or dyadic > 2 or Mc < 1 or rows > 64 * (4 >> min(dyadic, 2)) or (kind == 1 and not float(param) > 0):
            return None
        Mrows, Ncp = 256, ((Mc if kind == 0 else M) + 15) // 16 * 16
        P = A * (A + 1) // 2
        if P > self.SYM_TABLE_MAX_PAIRS:      # (the pair table is 8 bytes per pair: beyond 0.5 GiB the caller's blocked triangle serves)
            return None
        out = torch.empty(A, A, dtype=X.dtype, device=X.device)
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        lib = load()
        with _device(X.device):
            # one allocation, ONE staging launch (sk_prep_cat_*): rows | columns | the triangle's pairs [P][2] int32 right behind them
            nr, nt = A * Mrows * 8, A * 8 * Ncp
            buf = torch.empty(nr + nt + P, dtype=torch.float64, device=X.device)
            Xr, Xt, tab = buf[:nr], buf[nr:nr + nt], buf[nr + nt:]
            if kind == 0:
                s_rows, diff = float(lib.sk_linear_prescale(int(dyadic))) * float(param) ** 2, 1
            else:
                s_rows, diff = 1.0, 0
            _check(getattr(lib, "sk_prep_cat_" + _suffix(X))(_ptr(X), A, None, 0, M, D, diff, s_rows, 1.0, _ptr(Xr), None, Mrows, _ptr(Xt), Ncp, 8,
                                                             _ptr(tab), -1, _stream(X)), "sk_prep_cat")
            if kind == 0:
                rc = getattr(lib, "sk_solve_fwd_linear_sym_" + _suffix(X))(_ptr(Xr), _ptr(Xt), _ptr(tab), A, Mrows, Mc, Mc, Ncp, D, int(dyadic),
                                                                           scheme, _ptr(out), _ptr(_queue(X.device)), _stream(X))
            else:
                rc = getattr(lib, "sk_solve_fwd_rbf_sym_" + _suffix(X))(_ptr(Xr), _ptr(Xt), _ptr(tab), A, Mrows, Mc, Mc, Ncp, D, int(dyadic), scheme,
                                                                        1.0 / float(param), _ptr(out), _ptr(_queue(X.device)), _stream(X))
        if rc == 2:
            return None
        _check(rc, "sk_solve_fwd_*_sym")
        return out

    @staticmethod
    def _adjoint_mb_layout(P, Mc, Nc, dyadic, D, kind=1):
        """(mrows, rows, outw, edge_doubles, workspace_bytes, ncols) of sk_rbf_adjoint_fused_mb_f64 (kind 1) / sk_linear_adjoint_fused_mb_f64
        (kind 0; ncols = 0), or None outside its scope."""
        mrows, rows, outw, ncols = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        ed, wsb = ctypes.c_int64(0), ctypes.c_size_t(0)
        if kind == 0:
            rc = load().sk_linear_adjoint_fused_mb_layout(P, Mc, Nc, int(dyadic), D, ctypes.byref(mrows), ctypes.byref(rows), ctypes.byref(outw),
                                                          ctypes.byref(ed), ctypes.byref(wsb))
            return None if rc != 0 else (mrows.value, rows.value, outw.value, int(ed.value), int(wsb.value), 0)
        rc = load().sk_rbf_adjoint_fused_mb_layout(P, Mc, Nc, int(dyadic), D, ctypes.byref(mrows), ctypes.byref(rows), ctypes.byref(outw),
                                                   ctypes.byref(ncols), ctypes.byref(ed), ctypes.byref(wsb))
        if rc != 0:
            return None
        return mrows.value, rows.value, outw.value, int(ed.value), int(wsb.value), ncols.value

    def solve_fwd_fused_static(self, kind, param, X, Y, dyadic, naive, gram, swap=False, keep_edges=False):
        """K[MM][NN] with the static kernel (kind 0 linear / param = scale, 1 rbf / param = sigma) formed inside the solver, for
        pairs that need several bands of a wavefront and for path dims up to 16 (sk_solve_fwd_static_*, csrc/sk_wave_fused_mb.hip):
        nothing of size pairs x M x N in HBM; either scheme, any M and N.  None outside the kernel's scope (dyadic > 2, dim > 16).
        swap: solved as K(y, x) -- the kernel is symmetric in its arguments (both static kernels and the stencil are); what
        sk_route_query asks for when the second path is the short one.  keep_edges: (K, edges or None)."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 16 or not 0 <= dyadic <= 2 or Mc < 1 or Nc < 1 or (kind == 1 and not float(param) > 0):
            return None
        if swap:
            Kt = self.solve_fwd_fused_static(kind, param, Y, X, dyadic, naive, gram)
            Kt = None if Kt is None else (Kt.t().contiguous() if gram else Kt)
            return (Kt, None) if (keep_edges and Kt is not None) else Kt
        lib = load()
        P = A * B if gram else A
        nbytes = int(lib.sk_solve_fwd_static_workspace_bytes(int(kind), P, Mc, Nc, int(dyadic), D))
        if not nbytes:
            return None
        fd = 8 if D <= 8 else 16
        Mrows = int(lib.sk_solve_fwd_static_rows(int(kind), Mc, int(dyadic)))
        # keep_edges (RBF at dyadic 1..2, Linear): every band's bottom row and the terminal column of every pair, in the layout
        # sk_rbf_adjoint_fused_mb_f64 / sk_linear_adjoint_fused_mb_f64 read
        lay = self._adjoint_mb_layout(P, Mc, Nc, dyadic, D, int(kind)) if keep_edges else None
        edges = None
        if lay is not None:
            Mrows = lay[0]
            edges = torch.empty(P * lay[3], dtype=torch.float64, device=X.device)
        Ncp = int(lib.sk_solve_fwd_static_cols(int(kind), Nc))      # 2 NUp: short second paths are padded to the band pipeline's minimum
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        dev = X.device
        out = torch.empty((A, B) if gram else (A,), dtype=X.dtype, device=dev)
        with _device(dev):
            y32 = kind == 1 and fd == 16 and X.dtype == torch.float32 and dyadic >= 1
            if kind == 0:
                Xr = _prep_paths(X, True, False, float(param) ** 2, Mrows, fd)
                Yt = _prep_paths(Y, True, True, 1.0, Ncp, fd)
            else:
                Xr = _prep_paths(X, False, False, 1.0, Mrows, fd)
                if y32:      # fp32 points, two dimensions per 16-byte unit: half the LDS ring (sk_prep_paths_f32 layout 2)
                    Yt = torch.empty(B, fd // 2 + 1, Ncp, 2, dtype=torch.float32, device=dev)   # packed points + a row of fp64 norms
                    _check(lib.sk_prep_paths_f32(_ptr(Y), B, N, D, 0, 2, 1.0, _ptr(Yt), Ncp, fd, _stream(X)), "sk_prep_paths (packed fp32)")
                else:
                    Yt = _prep_paths(Y, False, True, 1.0, Ncp, fd)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if X.dtype == torch.float32:
                rc = lib.sk_solve_fwd_static_f32(int(kind), float(param), _ptr(Xr), _ptr(Yt), int(y32), A, B if gram else 0, Mrows, Mc,
                                                 Nc, Ncp, D, fd, int(dyadic), scheme, _ptr(out), _ptr(edges), _ptr(ws), nbytes,
                                                 _stream(X))
            else:
                rc = lib.sk_solve_fwd_static_f64(int(kind), float(param), _ptr(Xr), _ptr(Yt), A, B if gram else 0, Mrows, Mc, Nc, Ncp,
                                                 D, fd, int(dyadic), scheme, _ptr(out), _ptr(edges), _ptr(ws), nbytes, _stream(X))
        if rc == 2:
            return None
        _check(rc, "sk_solve_fwd_static")
        # split mode's status word (the last 8 bytes of the workspace: items whose bounded wait gave up -> NaN pairs), read on demand
        self.last_split_status = _SplitStatus(ws) if int(lib.sk_solve_fwd_static_split(int(kind), P, Mc, Nc, int(dyadic), D)) and not keep_edges and not y32 else None
        return (out, edges) if keep_edges else out

    FUSED_RESCUE_BLOCKS_MB = 8   # (a stored pair of 2044 x 2044 grids is 67 MB)

    def linear_adjoint_fused_mb(self, X, Y, param, dyadic, edges, scale, gram=True, kfinal=None, naive=False):
        """(dL/dX (A,M,D), worst self-check residual, reduced on demand: `float(r)`) for the LINEAR static kernel on LONG or WIDE paths
        straight from the paths and the edges solve_fwd_fused_static(0, ..., keep_edges=True) kept (sk_linear_adjoint_fused_mb_f64; fp64
        sweep; dim <= 16, dyadic 0..2, any M, N >= ~160).  None outside that scope."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 16 or dyadic not in (0, 1, 2) or Mc < 1 or Nc < 1 or A == 0 or B == 0 or edges is None:
            return None
        P, Bk = (A * B, B) if gram else (A, 0)
        lay = self._adjoint_mb_layout(P, Mc, Nc, dyadic, D, 0)
        if lay is None or edges.numel() != P * lay[3]:
            return None
        mrows, rows, fd, _, nbytes, _ = lay
        Ncp = int(load().sk_solve_fwd_static_cols(0, Nc))
        dev = X.device
        if scale is not None:
            scale = scale.double().contiguous()
        with _device(dev):
            dXr = _prep_paths(X, True, False, float(param) ** 2, mrows, fd)
            dYt = _prep_paths(Y, True, True, 1.0, Ncp, fd)
            tpart = torch.empty(P, rows, fd, dtype=torch.float64, device=dev)
            err = torch.zeros(P, dtype=torch.float64, device=dev)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            kf, rws, rws_bytes = self._fused_rescue_args(0, kfinal, P, Mc, Nc, dyadic, dev, self.FUSED_RESCUE_BLOCKS_MB)
            rc = load().sk_linear_adjoint_fused_mb_f64(_ptr(dXr), _ptr(dYt), A, Bk, mrows, Mc, Nc, Ncp, D, fd, int(dyadic),
                                                       SCHEME_NAIVE if naive else SCHEME_DEFAULT,
                                                       _ptr(edges), _ptr(scale), _ptr(tpart), tpart.numel(), _ptr(err), _ptr(ws), nbytes, _ptr(kf),
                                                       float(self.FUSED_SCREEN), float(self.ADJ_RESIDUAL_TOL), _ptr(rws), rws_bytes, _stream(X))
            if rc == 2:
                return None
            _check(rc, "sk_linear_adjoint_fused_mb")
        self.last_fused_err = err
        T = tpart.view(A, B if gram else 1, rows, fd).sum(1).flip(1)[:, :Mc, :D]    # the pairs of an a in a fixed order; flipped rows back to p
        g = torch.zeros(A, M, D, dtype=torch.float64, device=dev)
        g[:, 1:] += T          # d inc[p,q] / d x[p+1] = +s^2 dy[q]
        g[:, :-1] -= T         # d inc[p,q] / d x[p]   = -s^2 dy[q]
        if float(param) != 1.0:
            g = g * (float(param) ** 2)
        return g.to(X.dtype), _WorstResidual(err)

    def rbf_adjoint_fused_mb(self, X, Y, sigma, dyadic, edges, scale, gram=True, kfinal=None, naive=False):
        """(dL/dX (A,M,D), worst self-check residual, reduced on demand: `float(r)`) for the RBF static kernel on LONG or WIDE paths
        straight from the paths and the terminal edges solve_fwd_fused_static(keep_edges=True) kept: adjoint PDE, node evaluation and
        chain rule in one multi-band kernel (sk_rbf_adjoint_fused_mb_f64; fp64 sweep whatever the dtype of X; dim <= 16, dyadic
        1..2, any M, N >= ~160).  None outside that scope."""
        _dev(X, "X")
        _dev(Y, "Y")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 16 or dyadic not in (0, 1, 2) or Mc < 1 or Nc < 1 or A == 0 or B == 0 or not float(sigma) > 0 or edges is None:
            return None
        P, Bk = (A * B, B) if gram else (A, 0)
        lay = self._adjoint_mb_layout(P, Mc, Nc, dyadic, D)
        if lay is None or edges.numel() != P * lay[3]:
            return None
        mrows, rows, outw, _, nbytes, ncols = lay
        fd = outw - 2
        Ncp = ncols
        dev = X.device
        if scale is not None:
            scale = scale.double().contiguous()
        with _device(dev):
            Xr = _prep_paths(X, False, False, 1.0, mrows, fd)
            y32 = fd == 16 and X.dtype == torch.float32 and dyadic >= 1
            if y32:      # fp32 points, two dimensions per 16-byte unit + a row of fp64 norms: half the LDS ring (as the forward)
                Yt = torch.empty(B, fd // 2 + 1, Ncp, 2, dtype=torch.float32, device=dev)
                _check(load().sk_prep_paths_f32(_ptr(Y), B, N, D, 0, 2, 1.0, _ptr(Yt), Ncp, fd, _stream(X)), "sk_prep_paths (packed fp32)")
            else:
                Yt = _prep_paths(Y, False, True, 1.0, Ncp, fd)
            gpart = torch.empty(P, rows, outw, dtype=torch.float64, device=dev)
            gpart[:, 0].zero_()                      # node row 0: only the rescue writes there (the sweep's share comes through n0)
            n0 = torch.empty(P, ncols, dtype=torch.float64, device=dev)
            err = torch.zeros(P, dtype=torch.float64, device=dev)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            kf, rws, rws_bytes = self._fused_rescue_args(1, kfinal, P, Mc, Nc, dyadic, dev, self.FUSED_RESCUE_BLOCKS_MB)
            Yt64 = None if rws is None else (_prep_paths(Y, False, True, 1.0, Ncp, fd) if y32 else Yt)
            rc = load().sk_rbf_adjoint_fused_mb_f64(_ptr(Xr), _ptr(Yt), int(y32), A, Bk, mrows, Mc, Nc, Ncp, D, fd, int(dyadic),
                                                    SCHEME_NAIVE if naive else SCHEME_DEFAULT,
                                                    float(sigma), _ptr(edges), _ptr(scale), _ptr(gpart), gpart.numel(), _ptr(n0), n0.numel(),
                                                    _ptr(err), _ptr(ws), nbytes, _ptr(Yt64), _ptr(kf), float(self.FUSED_SCREEN),
                                                    float(self.ADJ_RESIDUAL_TOL), _ptr(rws), rws_bytes, _stream(X))
            if rc == 2:
                return None
            _check(rc, "sk_rbf_adjoint_fused_mb")
        self.last_fused_err = err
        T = gpart.view(A, B if gram else 1, rows, outw)[:, :, :M].sum(1)     # the pairs of an a added in a fixed order
        cs, accd = T[..., 0:1].clone(), T[..., 2:2 + D].clone()
        # node row 0: per-column weights, contracted with the points of y_b here
        n0v = n0.view(A, B if gram else 1, ncols)[:, :, :N]
        cs[:, 0, 0] += n0v.sum((1, 2))
        Yd = Y.double()
        # (one small product per y_b, then the sum over b: the flat (A x BN) (BN x D) product runs at 75 GFLOP/s in rocBLAS)
        accd[:, 0] += torch.bmm(n0v.transpose(0, 1), Yd).sum(0) if gram else torch.einsum("ac,acd->ad", n0v[:, 0], Yd)
        g = (-2.0 / float(sigma)) * (X.double() * cs - accd)               # sum_c V G (-2/sigma) (x_r - y_c)
        return g.to(X.dtype), _WorstResidual(err)

    def linear_adjoint_fused(self, X, Y, param, dyadic, edges, scale, gram=True, kfinal=None, naive=False, staged=None, gscale=None, yside=False):
        """(dL/dX (A,M,D), worst self-check residual, reduced on demand: `float(r)`) for the LINEAR static kernel straight from the paths
        and the forward's terminal edges: adjoint PDE and contraction in one kernel (sk_linear_adjoint_fused_f64; dim <= 8,
        dyadic <= 2, M - 1 <= 128 (64 at dyadic 2); computed in fp64 whatever the dtype of X).  None outside that scope.  The
        gradient is only valid when the residual is <= ADJ_RESIDUAL_TOL (not NaN) -- unless `kfinal` (the forward values, one
        per pair) is given: then the library's device-side rescue (sk_adj_fused_rescue.hip) takes exploding pairs out of the
        sweep, solves them with stored grids and adds their exact share, and the gradient is valid as returned -- nothing for the
        host to check, no synchronisation.  gram=False: paired batch, Y [A,N,D], scale [A].
        yside (Gram): the SECOND-argument sums INSTEAD -- (None, residual, sums (A, B, N-1, D)): per pair and increment q of y_b,
        sum_p W[a,b,p,q] s^2 (x_a[p+1] - x_a[p]) = d k(x_a, y_b) / d (y_b[q+1] - y_b[q]), WITHOUT the upstream gradient (`scale` is not
        used; see second_argument_gradient)."""
        _dev(X, "X")
        A, M, D = X.shape
        if staged is not None:      # (dXr [>= A][256][8] with s^2, dYt [B][8][Ncp], B, N): staged by the caller (loss_forward)
            dXr, dYt, B, N = staged
        else:
            _dev(Y, "Y")
            B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 8 or dyadic not in (0, 1, 2) or Mc < 1 or Nc < 1 or A == 0 or B == 0 or Mc > (64 if dyadic == 2 else 128):
            return None
        if yside and not gram:
            return None
        dev = X.device
        Mrows, Ncp = _stage_rows(0, M, gram), (Nc + 15) // 16 * 16
        if scale is not None:
            scale = scale.double().contiguous()
        lib = load()
        P, Bk = (A * B, B) if gram else (A, 0)
        ppg, rows, ycols = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        sizes = (ctypes.byref(ppg), ctypes.byref(rows), ctypes.byref(ycols) if yside else None)
        with _device(dev):
            if staged is None:
                # fp32 paths: differences of the up-cast points, as the edge-keeping forward forms them (both arrays in one launch)
                dXr, dYt = _prep_pair(X, Y, True, float(param) ** 2, Mrows, Ncp)
            args = (_ptr(dXr), _ptr(dYt), A, Bk, Mrows, Mc, Nc, Ncp, int(dyadic), SCHEME_NAIVE if naive else SCHEME_DEFAULT, _ptr(edges),
                    _ptr(scale))
            rc = lib.sk_linear_adjoint_fused_f64(*args, None, 0, None, None, 0, *sizes, None, 0.0, 0.0, None, 0, _stream(X))
            if rc == 2:
                return None
            _check(rc, "sk_linear_adjoint_fused (query)")
            chunks = -(-B // ppg.value) if gram else 1     # (chunk c of an a: [c B / chunks, (c + 1) B / chunks): ppg is the longest)
            self.last_fused_ppg = ppg.value      # (pairs per lane-group chunk of the last fused adjoint: what the tests look at)
            tpart = None if yside else torch.empty(A, chunks, rows.value, 8, dtype=torch.float64, device=dev)
            # every (pair, increment column < Nc) is written by the kernel; the padding columns up to ycols hold nothing meaningful
            ypart = torch.empty(A, B, ycols.value, 8, dtype=torch.float64, device=dev) if yside else None
            kf, rws, rws_bytes = self._fused_rescue_args(0, kfinal, P, Mc, Nc, dyadic, dev)
            # (with the rescue armed, its screening pass writes every residual entry before the sweep)
            err = torch.empty(P, dtype=torch.float64, device=dev) if kf is not None else torch.zeros(P, dtype=torch.float64, device=dev)
            rc = lib.sk_linear_adjoint_fused_f64(*args, _ptr(tpart), 0 if yside else tpart.numel(), _ptr(err), _ptr(ypart),
                                                 ypart.numel() if yside else 0, *sizes, _ptr(kf), float(self.FUSED_SCREEN),
                                                 float(self.ADJ_RESIDUAL_TOL), _ptr(rws), rws_bytes, _stream(X))
            if rc == 2:
                return None
            _check(rc, "sk_linear_adjoint_fused")
            if yside:
                self.last_fused_err = err
                return None, _WorstResidual(err), ypart[:, :, :Nc, :D]
        # worst self-check residual of the launch, NaN-propagating (torch.max does): stays on the device (diagnostics; with
        # `kfinal` the rescue has already dealt with exploding pairs: entries of -1 are pairs it took out of the sweep)
            # the chunks of an a added in ascending order, the flipped rows back to p, d inc[p,q] / d x[p+1] = +s^2 dy[q],
            # / d x[p] = -s^2 dy[q]: one launch (sk_linear_adjoint_finish_f64)
            g = torch.empty(A, M, D, dtype=torch.float64, device=dev)
            _check(lib.sk_linear_adjoint_finish_f64(_ptr(tpart), A, chunks, rows.value, M, D, float(param) ** 2, _ptr(gscale), _ptr(g), _stream(X)),
                   "sk_linear_adjoint_finish")
        res = _WorstResidual(err)
        self.last_fused_err = err
        if g.dtype != X.dtype:
            g = g.to(X.dtype)
        return g, res

    def rbf_adjoint_fused(self, X, Y, sigma, dyadic, edges, scale, gram=True, yside=False, kfinal=None, naive=False, staged=None, gscale=None,
                          yonly=False):
        """(dL/dX (A,M,D), worst self-check residual, reduced on demand: `float(r)`) for the RBF static kernel straight from the paths and
        the forward's terminal edges: adjoint PDE, node evaluation and chain rule in one kernel (sk_rbf_adjoint_fused_f64; fp64
        sweep whatever the dtype of X; dim <= 8, dyadic 1..2, one band per pair).  None outside that scope.  As for
        linear_adjoint_fused the gradient is valid only when the residual is <= ADJ_RESIDUAL_TOL, unless `kfinal` (forward values per
        pair) arms the device-side rescue.
        yside (Gram): a third result, the second-argument sums of the same sweep as a (A, B, N, 2 + D) tensor [S0, 0, S1]
        per node of y_b, WITHOUT the upstream gradient: d k(x_a, y_b) / d y_b[c] = (-2 / sigma) (y_b[c] S0 - S1) (see
        second_argument_gradient).  yonly, and always for paths of dim 5..8: the sums INSTEAD of the first-argument gradient (the first
        result is None) -- all a swapped call needs, and the kernel variant of dims 5..8 has registers for one of the two."""
        _dev(X, "X")
        A, M, D = X.shape
        if staged is not None:      # (Xr [>= A][256][8], Yt [B][8][Ncp], B, N): staged by the caller (loss_forward)
            Xr, Yt, B, N = staged
        else:
            _dev(Y, "Y")
            B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if D > 8 or dyadic not in (0, 1, 2) or (dyadic == 0 and (naive or M > 128)) or Mc < 1 or Nc < 1 or A == 0 or B == 0 or not float(sigma) > 0:
            return None
        if yside and not gram:
            return None
        yonly = yside and (yonly or D > 4)
        dev = X.device
        Mrows, Ncp = _stage_rows(1, M, gram), (N + 15) // 16 * 16
        if scale is not None:
            scale = scale.double().contiguous()
        lib = load()
        P, Bk = (A * B, B) if gram else (A, 0)
        ppg, rows, outw, ycols = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        with _device(dev):
            if staged is None:
                Xr, Yt = _prep_pair(X, Y, False, 1.0, Mrows, Ncp)
            args = (_ptr(Xr), _ptr(Yt), A, Bk, Mrows, Mc, Nc, Ncp, D, int(dyadic), SCHEME_NAIVE if naive else SCHEME_DEFAULT, float(sigma),
                    _ptr(edges), _ptr(scale))
            head = (ctypes.byref(ppg), ctypes.byref(rows), ctypes.byref(outw), ctypes.byref(ycols) if yside else None)
            tail = head + (None, 0.0, 0.0, None, 0, _stream(X))
            rc = lib.sk_rbf_adjoint_fused_f64(*args, None, 0, None, None, 0, *tail)
            if rc == 2:
                return None
            _check(rc, "sk_rbf_adjoint_fused (query)")
            chunks = -(-B // ppg.value) if gram else 1
            self.last_fused_ppg = ppg.value
            gpart = None if yonly else torch.empty(A, chunks, rows.value, outw.value, dtype=torch.float64, device=dev)
            # every (pair, node column < N) is written by the kernel; the padding columns up to ycols are not, and are never read
            ypart = torch.empty(A, B, ycols.value, 6 if D <= 4 else 10, dtype=torch.float64, device=dev) if yside else None
            kf, rws, rws_bytes = self._fused_rescue_args(1, kfinal, P, Mc, Nc, dyadic, dev)
            # (with the rescue armed, its screening pass writes every residual entry before the sweep)
            err = torch.empty(P, dtype=torch.float64, device=dev) if kf is not None else torch.zeros(P, dtype=torch.float64, device=dev)
            tail = head + (_ptr(kf), float(self.FUSED_SCREEN), float(self.ADJ_RESIDUAL_TOL), _ptr(rws), rws_bytes, _stream(X))
            rc = lib.sk_rbf_adjoint_fused_f64(*args, _ptr(gpart), 0 if yonly else gpart.numel(), _ptr(err), _ptr(ypart),
                                              ypart.numel() if yside else 0, *tail)
            if rc == 2:
                return None
            _check(rc, "sk_rbf_adjoint_fused")
            if yonly:
                self.last_fused_err = err
                return None, _WorstResidual(err), ypart[:, :, :N, :2 + D]
            # chunks of an a added in ascending order, then sum_c V G (-2/sigma) (x_r - y_c): one launch (sk_rbf_adjoint_finish_f64)
            X64 = X if X.dtype == torch.float64 else X.double()
            g = torch.empty(A, M, D, dtype=torch.float64, device=dev)
            _check(lib.sk_rbf_adjoint_finish_f64(_ptr(gpart), A, chunks, rows.value, outw.value, _ptr(X64), M, D, float(sigma), _ptr(gscale), _ptr(g),
                                                 _stream(X)), "sk_rbf_adjoint_finish")
        res = _WorstResidual(err)
        self.last_fused_err = err
        if g.dtype != X.dtype:
            g = g.to(X.dtype)
        if yside:
            return g, res, ypart[:, :, :N, :2 + D]
        return g, res

    @staticmethod
    def second_argument_gradient(ysums, Y, sigma, weight, b0=0):
        """dL/dY[b0:] (B-b0, N, D) from rbf_adjoint_fused(..., yside=True)'s sums (A, B, N, 2+D) and the per-pair upstream gradient
        `weight` (A, B): sum_a weight[a, b] (-2/sigma) (y_b[c] S0[a, b, c] - S1[a, b, c]).  The pairs are folded over a in a fixed
        order (one matrix product per b), so the result is reproducible.
        sigma None: from linear_adjoint_fused(..., yside=True)'s sums (A, B, N-1, D) per INCREMENT of y_b -- folded the same way, then
        differenced along the path (d inc[q] / d y[q+1] = +1, / d y[q] = -1)."""
        D = Y.shape[2]
        w = weight[:, b0:].double()
        ys = ysums[:, b0:]
        folded = torch.einsum("ab,abck->bck", w, ys)              # (B-b0, N, 2+D) / linear: (B-b0, N-1, D)
        if sigma is None:
            g = torch.zeros(folded.shape[0], Y.shape[1], D, dtype=torch.float64, device=Y.device)
            g[:, 1:] = folded
            g[:, :-1] -= folded
            return g.to(Y.dtype)
        g = (-2.0 / float(sigma)) * (Y[b0:].double() * folded[..., 0:1] - folded[..., 2:2 + D])
        return g.to(Y.dtype)

    FUSED_SCREEN = 1e3        # |K[MM][NN]| above which a pair leaves the fused adjoint's sweep for the stored-grid rescue
    FUSED_RESCUE_BLOCKS = 64  # flagged chunks re-solved concurrently

    def _fused_rescue_args(self, kind, kfinal, P, Mc, Nc, dyadic, dev, blocks=None):
        """(kfinal as fp64 [P] or None, workspace tensor or None, its bytes) for the fused adjoints' device-side rescue; no rescue
        (None, None, 0) without forward values or for grids the stored-grid kernel cannot hold."""
        if kfinal is None:
            return None, None, 0
        blocks = int(blocks or self.FUSED_RESCUE_BLOCKS)
        q = load().sk_fused_rescue_workspace_bytes
        nbytes = int(q(int(kind), P, Mc, Nc, int(dyadic), blocks))
        # the workspace holds `blocks` pairs of stored fine grids: bounded like the unfused route's scratch (a long second path
        # would otherwise ask for tens of GB per backward call); fewer blocks only make the -- rare -- rescue less concurrent
        while nbytes > self.GRID_SCRATCH_BYTES and blocks > 1:
            blocks = max(1, blocks // 2)
            nbytes = int(q(int(kind), P, Mc, Nc, int(dyadic), blocks))
        if not nbytes:
            return None, None, 0
        kf = kfinal.detach().reshape(-1).double().contiguous()
        if kf.numel() != P:
            raise ValueError("kfinal must hold one forward value per pair")
        return kf, torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes

    def static_adjoint(self, kind, param, X, Y, W, scale, gram):
        """dL/dX (A,M,D) from W = dL/d inc_c and the per-pair upstream gradient `scale`, for the fused static kernels
        (adjoint of static_increments; neither G_static nor dL/dG_static is materialised)."""
        _dev(X, "X")
        _dev(Y, "Y")
        W, ldw = _row_stride(W, "W")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        if scale is not None:
            _dev(scale, "scale")
            if scale.dtype != W.dtype:
                raise ValueError("scale must have W's dtype")
        with _device(X.device):
            fn = getattr(load(), "sk_static_adjoint_" + _suffix(X))
            if kind == 0:
                if D <= 8:      # pre-differenced, dimension-major y: every load of the contraction is coalesced
                    T = torch.empty(A, M - 1, D, dtype=X.dtype, device=X.device)
                    ldy = _padded_ld(N - 1, 8)
                    dYt = _prep_paths(Y, True, True, 1.0, ldy)
                    fl = getattr(load(), "sk_linear_adjoint_" + _suffix(X))
                    _check(fl(_ptr(dYt), ldy, _ptr(W), ldw, _ptr(scale), A, B if gram else 0, M - 1, N - 1, D, _ptr(T), _stream(X)),
                           "sk_linear_adjoint")
                elif D <= 32 and N <= 128:
                    # 9..32 dims: k_static_linear_adj_tiled reads W once, y_b through LDS, pairs loaded a chunk ahead (sk_static.hip)
                    T = torch.empty(A, M - 1, D, dtype=X.dtype, device=X.device)
                    _check(fn(0, float(param), _ptr(X), _ptr(Y), _ptr(W), ldw, _ptr(scale), A, B if gram else 0, M, N, D, _ptr(T),
                              _stream(X)), "sk_static_adjoint")
                else:
                    # wider or longer paths: T[a] = sum_b scale[a, b] W[a, b] (Mc x Nc) @ dY[b] (Nc x D) IS a batched matrix product --
                    # a plain library GEMM (rocBLAS under torch.matmul)
                    # (256 x 256 pairs of 100 points: dim 12 / 20 / 32 7.2 / 24.4 / 42.9 -> 2.3 / 2.5 / 4.9 ms, equal to 1e-15)
                    dY = Y[:, 1:] - Y[:, :-1]
                    Wv = W[..., :N - 1]
                    if gram:
                        # in column blocks of Y, so that the (A, b, Mc, D) products stay below ~256 MB whatever B is (for D
                        # close to N the un-blocked product was up to twice the size of W, outside the caller's tile accounting)
                        step = max(1, int((256 << 20) // max(1, A * (M - 1) * D * X.element_size())))
                        T = None
                        for b0 in range(0, B, step):
                            Tm = torch.matmul(Wv[:, b0:b0 + step], dY[b0:b0 + step])    # (A, b, Mc, D)
                            if scale is not None:
                                Tm = Tm.mul_(scale[:, b0:b0 + step, None, None])
                            T = Tm.sum(1) if T is None else T.add_(Tm.sum(1))
                            del Tm
                    else:
                        T = torch.matmul(Wv, dY)                                    # (A, Mc, D)
                        if scale is not None:
                            T = T * scale[:, None, None]
                g = torch.zeros(A, M, D, dtype=X.dtype, device=X.device)
                g[:, 1:] += T          # d inc[p,q] / d x[p+1] = +s^2 dy[q]
                g[:, :-1] -= T         # d inc[p,q] / d x[p]   = -s^2 dy[q]
                return g * (float(param) ** 2) if float(param) != 1.0 else g
            g = torch.empty(A, M, D, dtype=X.dtype, device=X.device)
            _check(fn(1, float(param), _ptr(X), _ptr(Y), _ptr(W), ldw, _ptr(scale), A, B if gram else 0, M, N, D, _ptr(g),
                      _stream(X)), "sk_static_adjoint")
            return g

    def static_adjoint2(self, kind, param, X, Y, W, scale, b0=0):
        """dL/dY[b0:] (B-b0, N, D) from W = dL/d inc_c of the Gram pairs (a, b) and the per-pair upstream gradient `scale`
        (A, B): the second-argument counterpart of static_adjoint (Gram only).  None outside its scope (linear: D > 8)."""
        _dev(X, "X")
        _dev(Y, "Y")
        W, ldw = _row_stride(W, "W")
        A, M, D = X.shape
        B, N = Y.shape[0], Y.shape[1]
        if (kind == 0 and D > 8) or D > self.MAX_FUSED_DIM:
            return None
        if scale is not None:
            _dev(scale, "scale")
            if scale.dtype != W.dtype or scale.numel() != A * B:
                raise ValueError("scale must be (A, B) with W's dtype")
        dev = X.device
        with _device(dev):
            fn = getattr(load(), "sk_static_adjoint2_" + _suffix(X))
            if kind == 0:
                dXr = _prep_paths(X, True, False, float(param) ** 2, M - 1)
                T2 = torch.empty(B - b0, N - 1, D, dtype=X.dtype, device=dev)
                _check(fn(0, float(param), None, None, _ptr(dXr), M - 1, _ptr(W), ldw, _ptr(scale), A, B, int(b0), M, N, D,
                          _ptr(T2), _stream(X)), "sk_static_adjoint2")
                g = torch.zeros(B - b0, N, D, dtype=X.dtype, device=dev)
                g[:, 1:] += T2
                g[:, :-1] -= T2
                return g
            g = torch.empty(B - b0, N, D, dtype=X.dtype, device=dev)
            _check(fn(1, float(param), _ptr(X), _ptr(Y), None, 0, _ptr(W), ldw, _ptr(scale), A, B, int(b0), M, N, D, _ptr(g),
                      _stream(X)), "sk_static_adjoint2")
            return g

    def increments_adjoint(self, W, scale=None):
        """W [..., M-1, N-1] (+ per-pair scale [...]) -> dG [..., M, N]."""
        W, ldw = _row_stride(W, "W")
        Mc, Nc = W.shape[-2:]
        P = W.numel() // (Mc * Nc)
        if scale is not None:
            _dev(scale, "scale")
            if scale.dtype != W.dtype or scale.numel() != P:
                raise ValueError("scale must have one entry per pair and W's dtype")
        out = torch.empty(W.shape[:-2] + (Mc + 1, Nc + 1), dtype=W.dtype, device=W.device)
        with _device(W.device):
            fn = getattr(load(), "sk_increments_adjoint_" + _suffix(W))
            _check(fn(_ptr(W), ldw, _ptr(scale), P, Mc + 1, Nc + 1, _ptr(out), _stream(W)), "sk_increments_adjoint")
        return out

    def solve_fwd(self, inc_c, dyadic, naive=False, flags=0, want_grid=False, want_edges=False):
        """inc_c [..., Mc, Nc] -> final [...]; optionally (grid [..., MM+1, NN+1], edges [..., MM+NN+2])."""
        if dyadic > 3 and not (want_grid or want_edges) and not (flags & (FLAG_SIMPLE | FLAG_EXACT)):
            # the streaming solver is built for dyadic orders 0..3: above that, dyadic 3 on refined increments (see _refined)
            Mc, Nc = inc_c.shape[-2:]
            batch = inc_c.shape[:-2]
            flat = inc_c.reshape(-1, Mc, Nc)
            out = torch.empty(flat.shape[0], dtype=inc_c.dtype, device=inc_c.device)
            for p0, p1, buf in self._refined(flat, 1 << (int(dyadic) - 3)):
                out[p0:p1] = self.solve_fwd(buf, 3, naive, flags)
            return out.reshape(batch)
        inc_c, ld = _row_stride(inc_c, "inc_c")
        Mc, Nc = inc_c.shape[-2:]
        batch = inc_c.shape[:-2]
        P = inc_c.numel() // (Mc * Nc)
        MM, NN = Mc << dyadic, Nc << dyadic
        out = torch.empty(batch, dtype=inc_c.dtype, device=inc_c.device)
        grid = torch.empty(batch + (MM + 1, NN + 1), dtype=inc_c.dtype, device=inc_c.device) if want_grid else None
        edges = torch.empty(batch + (MM + NN + 2,), dtype=torch.float64, device=inc_c.device) if want_edges else None
        with _device(inc_c.device):
            fn = getattr(load(), "sk_solve_fwd_" + _suffix(inc_c))
            _check(fn(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), SCHEME_NAIVE if naive else SCHEME_DEFAULT, int(flags),
                      _ptr(out), _ptr(grid), _ptr(edges), _stream(inc_c)), "sk_solve_fwd")
        if want_grid or want_edges:
            return out, grid, edges
        return out

    def solve_fwd_keep_edges(self, inc_c, dyadic, naive=False):
        """Forward solve that keeps the terminal row/column of every pair for a later solve_adj(..., edges=...):
        (final [...], edges) -- edges is None when the strip kernels do not cover the shape or layout (then final comes
        from the ordinary forward solve and the adjoint will run its own forward sweep)."""
        inc_c, ld = _row_stride(inc_c, "inc_c")
        Mc, Nc = inc_c.shape[-2:]
        batch = inc_c.shape[:-2]
        P = inc_c.numel() // (Mc * Nc)
        lib = load()
        nbytes = int(lib.sk_strip_edges_bytes(P, Mc, Nc, int(dyadic), inc_c.element_size()))
        if nbytes and (ld * inc_c.element_size()) % 128 == 0:
            out = torch.empty(batch, dtype=inc_c.dtype, device=inc_c.device)
            edges = torch.empty(nbytes // 8, dtype=torch.float64, device=inc_c.device)
            with _device(inc_c.device):
                fn = getattr(lib, "sk_solve_fwd_edges_" + _suffix(inc_c))
                rc = fn(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), SCHEME_NAIVE if naive else SCHEME_DEFAULT, _ptr(out), _ptr(edges),
                        _stream(inc_c))
            if rc == SK_OK:
                return out, edges
            if rc != 2:
                _check(rc, "sk_solve_fwd_edges")
        return self.solve_fwd(inc_c, dyadic, naive), None

    # residual of the fast adjoint's self-check above which a pair is re-solved by the stored-grid kernel
    ADJ_RESIDUAL_TOL = 1e-8

    def solve_adj(self, inc_c, dyadic, naive=False, flags=0, return_residual=False, edges=None):
        """inc_c [..., Mc, Nc] -> (final [...], W [..., Mc, Nc] = d final / d inc_c).

        The fast kernel recomputes K backwards instead of storing it and reports a per-pair residual; pairs whose
        residual exceeds ADJ_RESIDUAL_TOL (K exploding beyond ~1e4) are re-solved by the stored-grid kernel.
        `edges` (from solve_fwd_keep_edges on the same increments) skips the forward sweep; `final` is then None."""
        if edges is not None:
            # edges kept by another kernel in its own layout (sk_solve_fwd_static_* keeps sk_rbf_adjoint_fused_mb_f64's): sweep forward here
            P_ = inc_c.numel() // (inc_c.shape[-2] * inc_c.shape[-1])
            if edges.numel() * 8 != int(load().sk_strip_edges_bytes(P_, inc_c.shape[-2], inc_c.shape[-1], int(dyadic), inc_c.element_size())):
                edges = None
        if dyadic > 2 and edges is None and not (flags & (FLAG_SIMPLE | FLAG_EXACT)):
            # the fast adjoint keeps two PDE states in registers and is built for dyadic orders 0..2; the stored-grid kernel is ~100x
            # slower (256 x 256 pairs of 100 points at dyadic 3: 2.6 s).  Dyadic order d on inc_c IS dyadic order 2 on inc_c
            # replicated 2^(d-2) x 2^(d-2) and scaled by 4^-(d-2) (the reference's own tile(), sigkernel.py:218, :364; powers of two:
            # the same fine grid bit for bit), and W folds back by summing the blocks
            return self._solve_adj_refined(inc_c, dyadic, naive, flags, return_residual)
        if (inc_c.dtype == torch.float32 and dyadic == 2 and edges is None and not (flags & (FLAG_SIMPLE | FLAG_EXACT))):
            # the fused adjoint has no fp32 variant at dyadic 2 (16-column blocks do not fit the register file) and the
            # stored-grid kernel is ~100x slower: run the fp64 kernel on up-cast increments, in pair chunks of bounded size
            return self._solve_adj_upcast(inc_c, dyadic, naive, flags, return_residual)
        inc_c, ld = _row_stride(inc_c, "inc_c")
        Mc, Nc = inc_c.shape[-2:]
        batch = inc_c.shape[:-2]
        P = inc_c.numel() // (Mc * Nc)
        dev = inc_c.device
        out = torch.empty(batch, dtype=inc_c.dtype, device=dev)
        ldw = _padded_ld(Nc, inc_c.element_size())
        Wp = torch.empty(batch + (Mc, ldw), dtype=inc_c.dtype, device=dev)
        err = torch.empty(batch, dtype=torch.float64, device=dev)
        lib = load()
        scheme = SCHEME_NAIVE if naive else SCHEME_DEFAULT
        es = inc_c.element_size()
        with _device(dev):
            fn = getattr(lib, "sk_solve_adj_" + _suffix(inc_c))
            fast = False
            if edges is not None:
                _dev(edges, "edges")
                _check(fn(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), scheme, int(flags) | FLAG_EDGES_GIVEN, None, _ptr(Wp), ldw,
                          _ptr(err), _ptr(edges), edges.numel() * 8, _stream(inc_c)), "sk_solve_adj (edges given)")
                out = None
                fast = True
            else:
                rc = 2
                if not (flags & (FLAG_SIMPLE | FLAG_EXACT)):
                    # first the fast kernels, with the workspace THEY need (P (edges + 1) doubles): the stored-grid figure is
                    # min(P, 1024) whole pairs of grids -- tens of GB for long paths -- and is only taken when it is used
                    nbytes = int(lib.sk_adj_workspace_bytes(P, Mc, Nc, int(dyadic), FLAG_FAST_ONLY, es))
                    if nbytes:
                        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                        rc = fn(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), scheme, int(flags) | FLAG_FAST_ONLY, _ptr(out), _ptr(Wp),
                                ldw, _ptr(err), _ptr(ws), nbytes, _stream(inc_c))
                        if rc != 2:
                            _check(rc, "sk_solve_adj")
                            fast = True
                    if rc == 2 and (flags & FLAG_FAST_ONLY):
                        _check(rc, "sk_solve_adj")
                if rc == 2:      # stored-grid kernel: as many scratch slots as the budget allows (its blocks grid-stride)
                    ws, nbytes = self._grid_scratch(P, Mc, Nc, dyadic, dev, 1024)
                    _check(fn(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), scheme, (int(flags) & ~FLAG_FAST_ONLY) | FLAG_SIMPLE,
                              _ptr(out), _ptr(Wp), ldw, _ptr(err), _ptr(ws), nbytes, _stream(inc_c)), "sk_solve_adj (stored grids)")
            if fast:
                # pairs whose self-check residual is too large (exploding kernels) are re-solved with stored grids by a
                # kernel that reads the residuals itself: nothing comes back to the host, the backward pass never synchronises
                # (the stored-grid kernel keeps three diagonals of the fine grid in LDS: sk_adj_rescue_slot_bytes is 0 for the
                # grids it cannot hold -- more than ~6800 fine rows -- and such pairs have no rescue: their residuals stay in
                # `err` for the caller, NaN-marked W would be worse than the fast kernel's value)
                ws2, nb2 = self._grid_scratch(P, Mc, Nc, dyadic, dev, self.RESCUE_SLOTS)
                if nb2:
                    fr = getattr(lib, "sk_adj_rescue_" + _suffix(inc_c))
                    rc = fr(_ptr(inc_c), ld, P, Mc, Nc, int(dyadic), scheme, _ptr(err), float(self.ADJ_RESIDUAL_TOL), _ptr(out),
                            _ptr(Wp), ldw, _ptr(ws2), nb2, _stream(inc_c))
                    if rc != 2:          # 2: the grid does not fit the stored-grid kernel's LDS -- no rescue available, not an error
                        _check(rc, "sk_adj_rescue")
        W = Wp[..., :Nc]
        # the caching allocator keeps the scratch alive for the work queued on this same stream
        if return_residual:
            return out, W, err
        return out, W

    RESCUE_SLOTS = 64                   # flagged pairs re-solved concurrently by sk_adj_rescue_*
    GRID_SCRATCH_BYTES = 2 << 30        # cap on the stored-grid scratch (whole pairs of solution grids)

    def _grid_scratch(self, P, Mc, Nc, dyadic, dev, max_slots):
        """(uint8 tensor, bytes) holding up to max_slots pairs of solution grids, within GRID_SCRATCH_BYTES (at least one slot;
        raises when a single slot exceeds half of the free memory even after releasing torch's cached blocks; (None, 0) when
        the stored-grid kernel cannot hold the grid at all)."""
        if not int(load().sk_adj_rescue_slot_bytes(Mc, Nc, int(dyadic))):
            return None, 0
        slot = int(load().sk_adj_rescue_slot_bytes(Mc, Nc, int(dyadic)))
        n = max(1, min(int(max_slots), int(P), self.GRID_SCRATCH_BYTES // max(slot, 1)))
        free, _ = torch.cuda.mem_get_info(dev)
        if n * slot > 0.5 * free:
            n = int(0.5 * free // slot)
        if n < 1:
            torch.cuda.empty_cache()      # blocks cached by torch's allocator do not show in mem_get_info: release them and look again
            free, _ = torch.cuda.mem_get_info(dev)
            n = int(0.5 * free // slot)
        if n < 1:
            raise RuntimeError("sigkernel_amd: no memory for one stored-grid scratch slot (%d bytes): pairs that fail the fast "
                               "adjoint's self-check could not be re-solved" % slot)
        return torch.empty(n * slot, dtype=torch.uint8, device=dev), n * slot

    def deriv_increments(self, G0, G1, G2, eps):
        """Static Gram matrices [..., M, N] of X, X + eps*gamma, X + 2*eps*gamma -> one tensor [3, ..., M-1, N-1]:
        increments of k and of its first / second finite-difference derivative along gamma (sigkernel.py:526-541)."""
        for t, n in ((G0, "G0"), (G1, "G1"), (G2, "G2")):
            _dev(t, n)
        if not (G0.shape == G1.shape == G2.shape and G0.dtype == G1.dtype == G2.dtype):
            raise ValueError("G0, G1, G2 must share shape and dtype")
        M, N = G0.shape[-2:]
        P = G0.numel() // (M * N)
        ld = _padded_ld(N - 1, G0.element_size())
        out = torch.empty((3,) + G0.shape[:-2] + (M - 1, ld), dtype=G0.dtype, device=G0.device)
        with _device(G0.device):
            fn = getattr(load(), "sk_deriv_increments_" + _suffix(G0))
            _check(fn(_ptr(G0), _ptr(G1), _ptr(G2), float(eps), P, M, N, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), ld,
                      _stream(G0)), "sk_deriv_increments")
        return out[..., : N - 1]

    def static_deriv_increments(self, kind, param, X0, X1, X2, Y, eps):
        """deriv_increments with the static kernel fused in (kind 0 = linear, 1 = rbf/sigma): X0 = X, X1 = X + eps*gamma,
        X2 = X + 2*eps*gamma (A,M,D), Y (B,N,D) -> [3, A, B, M-1, N-1]; None when D > 32 (generic route instead)."""
        for t, n in ((X0, "X0"), (X1, "X1"), (X2, "X2"), (Y, "Y")):
            _dev(t, n)
        A, M, D = X0.shape
        B, N = Y.shape[0], Y.shape[1]
        if D > self.MAX_FUSED_DIM or M < 2 or N < 2:
            return None
        ld = _padded_ld(N - 1, X0.element_size())
        out = torch.empty(3, A, B, M - 1, ld, dtype=X0.dtype, device=X0.device)
        with _device(X0.device):
            fn = getattr(load(), "sk_static_deriv_increments_" + _suffix(X0))
            _check(fn(int(kind), float(param), _ptr(X0), _ptr(X1), _ptr(X2), _ptr(Y), A, B, M, N, D, float(eps), _ptr(out[0]),
                      _ptr(out[1]), _ptr(out[2]), ld, _stream(X0)), "sk_static_deriv_increments")
        return out[..., : N - 1]

    def solve_deriv_fused(self, kind, param, X0, X1, X2, Y, dyadic, eps):
        """(k, d/dgamma k, d2/dgamma2 k), (A,B) each, straight from the paths X0 = X, X1 = X + eps*gamma, X2 = X + 2*eps*gamma and Y:
        the three increment arrays are formed inside the solver (sk_solve_deriv_static_f64, csrc/sk_wave_deriv_fused.hip) with the
        arithmetic of static_deriv_increments and never exist in HBM.  None outside its scope (fp64, dim <= 8, dyadic <= 2, second
        path of 126 points or more)."""
        for t, n in ((X0, "X0"), (X1, "X1"), (X2, "X2"), (Y, "Y")):
            _dev(t, n)
        A, M, D = X0.shape
        B, N = Y.shape[0], Y.shape[1]
        Mc, Nc = M - 1, N - 1
        if X0.dtype != torch.float64 or D > 8 or not 0 <= dyadic <= 2 or Mc < 1 or Nc < 1 or A == 0 or B == 0:
            return None
        if kind == 1 and not float(param) > 0:
            return None
        lib = load()
        mrows = ctypes.c_int(0)
        nbytes = int(lib.sk_solve_deriv_static_workspace_bytes(A * B, Mc, Nc, int(dyadic), D, ctypes.byref(mrows)))
        if not nbytes:
            return None
        fd = 8
        Ncp = 2 * (((Nc + 2) // 2 + 7) // 8 * 8)
        dev = X0.device
        out = torch.empty(3, A, B, dtype=torch.float64, device=dev)
        with _device(dev):
            Xr = [_prep_paths(x.contiguous(), False, False, 1.0, mrows.value, fd) for x in (X0, X1, X2)]
            Yt = _prep_paths(Y.contiguous(), False, True, 1.0, Ncp, fd)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.sk_solve_deriv_static_f64(int(kind), float(param), _ptr(Xr[0]), _ptr(Xr[1]), _ptr(Xr[2]), _ptr(Yt), A, B, mrows.value, Mc, Nc,
                                               Ncp, D, fd, int(dyadic), SCHEME_DEFAULT, float(eps), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                               _ptr(ws), nbytes, _stream(X0))
        if rc == 2:
            return None
        _check(rc, "sk_solve_deriv_static")
        return out[0], out[1], out[2]

    def solve_deriv(self, inc3, dyadic, flags=0):
        """inc3 [3, ..., Mc, Nc] (increments of k, d/dgamma, d2/dgamma2) -> (k, k_gamma, k_gamma_gamma), [...] each."""
        if inc3.shape[0] != 3:
            raise ValueError("inc3 must stack the three increment arrays on dim 0")
        if dyadic > 2 and not (flags & (FLAG_SIMPLE | FLAG_EXACT)):
            # the fast three-state solver is built for dyadic orders 0..2: above that, dyadic 2 on refined increments (the reference
            # tiles all three arrays alike, sigkernel.py:543-566)
            Mc, Nc = inc3.shape[-2:]
            batch = inc3.shape[1:-2]
            flat = inc3.reshape(3, -1, Mc, Nc)
            out = torch.empty(3, flat.shape[1], dtype=inc3.dtype, device=inc3.device)
            f = 1 << (int(dyadic) - 2)
            ldr = _padded_ld(Nc * f, 8)
            step = max(1, int(self._refined_chunk_bytes(flat) // (3 * Mc * f * ldr * 8)))
            for p0 in range(0, flat.shape[1], step):
                p1 = min(flat.shape[1], p0 + step)
                buf = torch.zeros(3, p1 - p0, Mc * f, ldr, dtype=torch.float64, device=inc3.device)
                self._fill_refined(buf, flat[:, p0:p1], f)
                k, kd, kdd = self.solve_deriv(buf[..., :Nc * f], 2, flags)
                out[0, p0:p1], out[1, p0:p1], out[2, p0:p1] = k, kd, kdd
                del buf
            return out[0].reshape(batch), out[1].reshape(batch), out[2].reshape(batch)
        if inc3.dtype == torch.float32 and dyadic == 2 and not (flags & (FLAG_SIMPLE | FLAG_EXACT)):
            # no fp32 fast kernel at dyadic 2 (register file): the fp64 one on up-cast increments, in bounded chunks
            Mc, Nc = inc3.shape[-2:]
            batch = inc3.shape[1:-2]
            P = inc3[0].numel() // (Mc * Nc)
            flat = inc3.reshape(3, P, Mc, Nc)
            ld64 = _padded_ld(Nc, 8)
            out = torch.empty(3, P, dtype=torch.float32, device=inc3.device)
            step = max(1, int(self.UPCAST_CHUNK_BYTES // (3 * Mc * ld64 * 8)))
            for p0 in range(0, P, step):
                p1 = min(P, p0 + step)
                buf = torch.zeros(3, p1 - p0, Mc, ld64, dtype=torch.float64, device=inc3.device)
                buf[..., :Nc] = flat[:, p0:p1]
                k, kd, kdd = self.solve_deriv(buf[..., :Nc], dyadic, flags)
                out[0, p0:p1], out[1, p0:p1], out[2, p0:p1] = k, kd, kdd
                del buf
            return out[0].reshape(batch), out[1].reshape(batch), out[2].reshape(batch)
        inc3, ld = _row_stride(inc3, "inc3")
        Mc, Nc = inc3.shape[-2:]
        batch = inc3.shape[1:-2]
        P = inc3[0].numel() // (Mc * Nc)
        out = torch.empty((3,) + batch, dtype=inc3.dtype, device=inc3.device)
        with _device(inc3.device):
            fn = getattr(load(), "sk_solve_deriv_" + _suffix(inc3))
            _check(fn(_ptr(inc3[0]), _ptr(inc3[1]), _ptr(inc3[2]), ld, P, Mc, Nc, int(dyadic), int(flags), _ptr(out[0]),
                      _ptr(out[1]), _ptr(out[2]), _stream(inc3)), "sk_solve_deriv")
        return out[0], out[1], out[2]

    UPCAST_CHUNK_BYTES = 4 << 30   # fp64 copy of the increments handled at a time by _solve_adj_upcast

    def _solve_adj_upcast(self, inc_c, dyadic, naive, flags, return_residual):
        Mc, Nc = inc_c.shape[-2:]
        batch = inc_c.shape[:-2]
        P = inc_c.numel() // (Mc * Nc)
        dev = inc_c.device
        flat = inc_c.reshape(P, Mc, Nc)
        ld64 = _padded_ld(Nc, 8)
        ldw = _padded_ld(Nc, 4)
        out = torch.empty(P, dtype=torch.float32, device=dev)
        Wp = torch.empty(P, Mc, ldw, dtype=torch.float32, device=dev)
        err = torch.empty(P, dtype=torch.float64, device=dev)
        step = max(1, int(self.UPCAST_CHUNK_BYTES // (Mc * ld64 * 8)))
        for p0 in range(0, P, step):
            p1 = min(P, p0 + step)
            buf = torch.zeros(p1 - p0, Mc, ld64, dtype=torch.float64, device=dev)
            buf[..., :Nc] = flat[p0:p1]
            o, W, e = self.solve_adj(buf[..., :Nc], dyadic, naive, flags, return_residual=True)
            out[p0:p1] = o
            Wp[p0:p1, :, :Nc] = W
            err[p0:p1] = e
            del buf, W
        res = (out.reshape(batch), Wp.reshape(batch + (Mc, ldw))[..., :Nc])
        return res + (err.reshape(batch),) if return_residual else res


    def _refined_chunk_bytes(self, src, chunk_bytes=None):
        """Bytes of refined increments handled at a time: UPCAST_CHUNK_BYTES at most, and no more than twice what the caller holds
        already (its tile of coarse increments, which it sized to ITS budget -- SigKernel(workspace_bytes=...)), 256 MiB at least."""
        cap = max(256 << 20, 2 * src.numel() * src.element_size())
        return min(int(chunk_bytes or self.UPCAST_CHUNK_BYTES), cap)

    @staticmethod
    def _fill_refined(buf, src, f):
        """buf [..., Mc f, ldr] (fp64, zeroed) <- src [..., Mc, Nc] replicated f x f and scaled by 1 / f^2, written through a strided view:
        no replicated temporary (repeat_interleave made two of the chunk's size)."""
        Mc, Nc = src.shape[-2:]
        lead = tuple(src.shape[:-2])
        ldr = buf.shape[-1]
        strides, acc = [], Mc * f * ldr
        for n in reversed(lead):
            strides.append(acc)
            acc *= n
        view = buf.as_strided(lead + (Mc, f, Nc, f), tuple(reversed(strides)) + (f * ldr, ldr, f, 1))
        view.copy_((src.double() * (1.0 / (f * f)))[..., :, None, :, None])

    def _refined(self, flat, f, chunk_bytes=None):
        """Chunks (p0, p1, refined) of flat [P, Mc, Nc]: every increment replicated f x f and scaled by 1 / f^2 (powers of two: exact), in
        fp64, rows zero-padded to whole 128-byte lines -- dyadic order d on flat is dyadic order d - log2 f on these, the same fine grid
        bit for bit (the reference's own tile(), sigkernel.py:218, :364)."""
        P, Mc, Nc = flat.shape
        ldr = _padded_ld(Nc * f, 8)
        step = max(1, int(self._refined_chunk_bytes(flat, chunk_bytes) // (Mc * f * ldr * 8)))
        for p0 in range(0, P, step):
            p1 = min(P, p0 + step)
            buf = torch.zeros(p1 - p0, Mc * f, ldr, dtype=torch.float64, device=flat.device)
            self._fill_refined(buf, flat[p0:p1], f)
            yield p0, p1, buf[..., :Nc * f]

    def _solve_adj_refined(self, inc_c, dyadic, naive, flags, return_residual):
        Mc, Nc = inc_c.shape[-2:]
        batch = inc_c.shape[:-2]
        P = inc_c.numel() // (Mc * Nc)
        dev = inc_c.device
        f = 1 << (int(dyadic) - 2)
        flat = inc_c.reshape(P, Mc, Nc)
        out = torch.empty(P, dtype=inc_c.dtype, device=dev)
        W = torch.empty(P, Mc, Nc, dtype=inc_c.dtype, device=dev)
        err = torch.empty(P, dtype=torch.float64, device=dev)
        scale = 1.0 / (f * f)
        for p0, p1, buf in self._refined(flat, f):
            o, Wr, e = self.solve_adj(buf, 2, naive, flags, return_residual=True)
            out[p0:p1] = o
            W[p0:p1] = (Wr.reshape(p1 - p0, Mc, f, Nc, f).sum(dim=(2, 4)) * scale).to(inc_c.dtype)
            err[p0:p1] = e
            del buf, Wr
        res = (out.reshape(batch), W.reshape(batch + (Mc, Nc)))
        return res + (err.reshape(batch),) if return_residual else res


_backend = HipBackend()


def get_backend():
    return _backend


def set_backend(b):
    """Swap the solver back-end (test seam: tests/ install an oracle-backed fake to exercise the
    host logic without a GPU). Returns the previous back-end."""
    global _backend
    prev, _backend = _backend, b
    return prev
