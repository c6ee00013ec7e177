"""The C-ABI library builds, loads and exports every symbol include/sigkernel_amd.h declares.
No compute calls: this runs without a GPU."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "sigkernel_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for n in ("sk_solve_fwd_f64", "sk_solve_fwd_f32", "sk_solve_adj_f64", "sk_solve_adj_f32", "sk_increments_f64",
              "sk_increments_adjoint_f64", "sk_adj_workspace_bytes", "sk_version", "sk_status_string"):
        assert n in names


def test_library_builds_and_exports_every_declared_symbol():
    from sigkernel_amd import build, _lib
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export: " + name
    # the ctypes signature table covers the header one to one
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.load().sk_version() >= 100
    assert _lib.load().sk_status_string(1).decode().startswith("bad argument")


def test_argument_errors_are_reported_without_a_device():
    from sigkernel_amd import _lib
    lib = _lib.load()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.sk_solve_fwd_f64(None, 0, 1, 4, 4, 0, 0, 0, None, None, None, None) == 1
    assert lib.sk_solve_fwd_f64(ctypes.c_void_p(16), 0, 1, 0, 4, 0, 0, 0, ctypes.c_void_p(16), None, None, None) == 1
    assert lib.sk_solve_fwd_f64(ctypes.c_void_p(16), 0, 1, 4, 4, 0, 7, 0, ctypes.c_void_p(16), None, None, None) == 1
    assert lib.sk_solve_fwd_f64(ctypes.c_void_p(16), 3, 1, 4, 4, 0, 0, 0, ctypes.c_void_p(16), None, None, None) == 1   # ld < Nc
    assert lib.sk_increments_f64(None, 1, 4, 4, None, 0, None) == 1
    assert lib.sk_solve_adj_f64(ctypes.c_void_p(16), 0, 1, 4, 4, 0, 0, 0, None, None, 0, None, None, 0, None) == 1
    p = ctypes.c_void_p(16)
    # EDGES_GIVEN (8) cannot be combined with the exact / simple kernels, and needs a workspace
    assert lib.sk_solve_adj_f64(p, 0, 1, 4, 4, 1, 0, 8 | 2, None, p, 0, p, p, 64, None) == 1
    assert lib.sk_solve_adj_f64(p, 16, 1, 4, 4, 1, 0, 8, None, p, 16, p, None, 0, None) == 4     # SK_ERR_WORKSPACE
    assert lib.sk_solve_fwd_edges_f64(p, 16, 1, 4, 4, 1, 0, None, p, None) == 1                # no output vector
    assert lib.sk_solve_fwd_edges_f64(p, 16, 1, 4, 4, 3, 0, p, p, None) == 2                   # dyadic 3: not covered
    assert lib.sk_strip_edges_bytes(10, 63, 63, 1, 8) == 10 * (128 + 128) * 8                  # NNp = 32 units * 4, MMp = 32 lanes * 4
    assert lib.sk_strip_edges_bytes(10, 63, 63, 3, 8) == 0 and lib.sk_strip_edges_bytes(10, 63, 63, 2, 4) == 0
    assert lib.sk_solve_deriv_f64(p, None, p, 0, 1, 4, 4, 0, 0, p, p, p, None) == 1
    assert lib.sk_deriv_increments_f64(p, p, p, 0.0, 1, 4, 4, p, p, p, 0, None) == 1              # eps must be positive
    assert lib.sk_linear_adjoint_f64(p, 2, p, 0, None, 1, 1, 4, 4, 2, p, None) == 1               # ldy < Nc
    assert lib.sk_solve_fwd_rbf_f64(p, p, 1, 1, 256, 4, 4, 16, 3, 1, 0, 0.0, p, None) == 1           # 1/sigma must be positive
    assert lib.sk_solve_fwd_rbf_f64(p, None, 1, 1, 256, 4, 4, 16, 3, 1, 0, 1.0, p, None) == 1
    assert lib.sk_solve_fwd_rbf_f32(p, p, 1, 1, 256, 4, 4, 16, 3, 3, 0, 1.0, p, None) == 2           # dyadic 3: not covered
    assert lib.sk_solve_fwd_rbf_f64(p, p, 1, 1, 256, 128, 4, 16, 3, 1, 0, 1.0, p, None) == 2      # 129 node rows: two bands
    assert lib.sk_solve_fwd_linear_f64(p, p, 1, 1, 256, 4, 4, 16, 0, 1, 0, p, None) == 1                 # path dimension 0
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, -1, 256, 4, 4, 16, 1, 0, p, None, None, 0, None, None, None, None) == 1  # B < 0
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, 2, 256, 4, 4, 16, 3, 0, p, None, None, 0, None, None, None, None) == 2   # dyadic 3


def test_product_path_fails_loudly_on_cpu_tensors():
    import pytest
    import torch
    import sigkernel_amd
    sk = sigkernel_amd.SigKernel(sigkernel_amd.LinearKernel(), 0)
    X = torch.rand(2, 4, 2, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.compute_Gram(X, X)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.compute_kernel(X, X)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sigkernel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libsk_oracle" not in src, f


def test_no_instruction_touches_an_in_flight_asynchronous_load():
    """sk_wave_adj.hip and sk_wave_deriv.hip issue loads whose wait is a separate inline-asm s_waitcnt; the compiler does not
    know the destination is still in flight, so the generated ISA is linted for any access in between."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_hazards.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_the_hazard_lint_recognises_a_hazard():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_async_hazards as lint
    bad = """_Zbad: ; @_Zbad
\tglobal_load_dwordx2 v[10:11], v[4:5], off
\tv_mov_b64_e32 v[2:3], v[10:11]
\ts_waitcnt vmcnt(0)
.Lfunc_end0:
"""
    good = bad.replace("\tv_mov_b64_e32 v[2:3], v[10:11]\n\ts_waitcnt vmcnt(0)", "\ts_waitcnt vmcnt(0)\n\tv_mov_b64_e32 v[2:3], v[10:11]")
    assert len(lint.scan(bad)) == 1 and len(lint.scan(good)) == 0
