"""The C-ABI library builds, loads and exports every symbol include/sigkernel_amd.h declares.
No compute calls: this runs without a GPU."""
import ctypes
import glob
import sys
import os
import re

import pytest

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
    assert lib.sk_solve_fwd_rbf_f64(p, p, 1, 1, 256, 4, 4, 16, 3, 1, 0, 0.0, p, None, None) == 1           # 1/sigma must be positive
    assert lib.sk_solve_fwd_rbf_f64(p, None, 1, 1, 256, 4, 4, 16, 3, 1, 0, 1.0, p, None, None) == 1
    assert lib.sk_solve_fwd_rbf_f32(p, p, 1, 1, 256, 4, 4, 16, 3, 3, 0, 1.0, p, None, None) == 2           # dyadic 3: not covered
    assert lib.sk_solve_fwd_rbf_f64(p, p, 1, 1, 256, 128, 4, 16, 3, 1, 0, 1.0, p, None, None) == 2      # 129 node rows: two bands
    assert lib.sk_solve_fwd_linear_f64(p, p, 1, 1, 256, 4, 4, 16, 0, 1, 0, p, None, None) == 1                 # path dimension 0
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, -1, 256, 4, 4, 16, 1, 0, p, None, None, 0, None, None, 0, None, None, None, None, 0.0, 0.0, None, 0, None) == 1  # B < 0
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, 2, 256, 4, 4, 16, 3, 0, p, None, None, 0, None, None, 0, None, None, None, None, 0.0, 0.0, None, 0, None) == 2   # dyadic 3
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, 2, 256, 4, 4, 16, 1, 0, p, None, p, 64, p, None, 0, None, None, None, p, 1e3, 1e-8, None, 0, None) == 1   # forward values without a rescue workspace
    assert lib.sk_linear_adjoint_fused_f64(p, p, 1, 2, 256, 4, 4, 16, 1, 0, p, None, None, 0, None, p, 64, None, None, None, None, 0.0, 0.0, None, 0, None) == 1   # second-argument sums without a residual array
    assert lib.sk_fused_rescue_workspace_bytes(1, 100, 63, 63, 2, 4) > 0 and lib.sk_fused_rescue_workspace_bytes(2, 100, 63, 63, 2, 4) == 0


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


def test_build_info_names_the_toolchain_and_the_sources():
    """sk_build_info(): the hipcc / clang versions and the hash of the sources this binary was built from -- the hand-scheduled kernels
    are linted against THAT compiler's register allocation at build time (csrc/Makefile), so a library must say what built it."""
    import hashlib
    import re
    from sigkernel_amd import _lib
    info = _lib.load().sk_build_info().decode()
    assert "gfx950" in info and "HIP version" in info and "clang version" in info and "hazard lint passed" in info
    m = re.search(r"sources ([0-9a-f]{16});", info)
    assert m, info
    csrc = os.path.join(ROOT, "sigkernel_amd", "csrc")
    files = sorted([os.path.join("..", "..", "include", "sigkernel_amd.h"), "Makefile"] +
                   [f for f in os.listdir(csrc) if f.endswith((".hip", ".h"))])
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    assert m.group(1) == h.hexdigest()[:16], "libsigkernel_amd.so is stale: rebuild (python -m sigkernel_amd.build)"
    # the gate itself: the build refuses to link when the lint finds a hazard
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "lint.ok" in mk and "check_async_hazards.py" in mk and "$(OUT): $(OBJS) $(OBJDIR)/sk_build_info.o $(OBJDIR)/lint.ok" in mk


def test_no_instruction_touches_an_in_flight_asynchronous_load():
    """sk_wave_adj.hip and sk_wave_deriv.hip issue loads whose wait is a separate inline-asm s_waitcnt; the compiler does not
    know the destination is still in flight, so the generated ISA is linted for any access in between."""
    import glob
    import subprocess
    import sys
    # the ISA the library was assembled from lies next to its objects (csrc/Makefile compiles with -save-temps and runs this lint as
    # a build gate); without it (a library built elsewhere) the lint compiles the units itself
    isa = sorted(glob.glob(os.path.join(ROOT, "sigkernel_amd", "csrc", "obj", "sk_wave*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "check_async_hazards.py")] + (["--asm"] + isa if len(isa) >= 8 else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
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


def _wave_shares(P, G, waves, resident, wpb=4, n_cu=256):
    import numpy as np
    from sigkernel_amd import _lib
    first, end, ppg = np.zeros(waves, np.int64), np.zeros(waves, np.int64), np.zeros(waves, np.int32)
    nr = _lib.load().sk_plan_wave_shares(P, G, waves, resident, wpb, n_cu, first.ctypes.data, end.ctypes.data, ppg.ctypes.data)
    return nr, first, end, ppg


@pytest.mark.parametrize("P,G,wpc", [(262144, 1, 12), (262144, 1, 8), (4194304, 1, 12), (2098176, 1, 12), (65536, 1, 8), (1048576, 4, 8),
                                     (100003, 2, 12), (12289, 1, 12), (3072 * 12 + 5, 1, 12), (3071, 1, 12), (1, 1, 12), (999983, 8, 16)])
def test_shares_by_age_rank_partition_the_pairs(P, G, wpc, monkeypatch):
    """sk_plan_wave_shares (the split the persistent kernels use): whatever the shares, every pair belongs to exactly
    one lane group -- group g of wave w sweeps [first + g ppg, first + (g + 1) ppg) clipped to its rank's end."""
    import numpy as np
    for weights in (None, "34,33,33", "80,15,5", "50,50", "97,3", "25,25,25,25", "1,1,98"):
        if weights is None:
            monkeypatch.delenv("SK_RANK_W", raising=False)
        else:
            monkeypatch.setenv("SK_RANK_W", weights)
        resident = 256 * wpc
        waves = min((P + G - 1) // G, resident)
        nr, first, end, ppg = _wave_shares(P, G, waves, resident)
        assert nr >= 1
        seen = np.zeros(P, np.int32)
        for w in range(waves):
            for g in range(G):
                lo = min(first[w] + g * ppg[w], end[w])
                hi = min(first[w] + (g + 1) * ppg[w], end[w])
                if hi > lo:
                    seen[lo:hi] += 1
        assert seen.min() == 1 and seen.max() == 1, (weights, nr)
        if nr > 1:      # the older the rank, the larger the share (default and the descending overrides)
            per_rank = [int(ppg[r * 1024]) for r in range(nr)]
            if weights in (None, "80,15,5", "97,3"):
                assert per_rank == sorted(per_rank, reverse=True) and per_rank[0] > per_rank[-1]


@pytest.mark.parametrize("A,B,max_groups,G", [(512, 512, 2048, 1), (1024, 2048, 2048, 1), (256, 2048, 2048, 1), (300, 2048, 2048, 1),
                                              (2048, 2048, 2048, 1), (64, 96, 3072, 1), (1024, 64, 8192, 4), (7, 5, 2048, 1), (100, 0, 2048, 1)])
def test_chunks_by_age_rank_partition_the_pairs(A, B, max_groups, G, monkeypatch):
    """sk_plan_group_chunks (the fused adjoints' split): the chunks of every row a tile its B pairs exactly once and the slots
    a * chunks + c are all distinct, whatever the shares."""
    import numpy as np
    from sigkernel_amd import _lib
    PPG = B if B > 0 else 1
    for d in range(1, max(B, 1) + 1):
        if B > 0 and B % d == 0 and A * (B // d) <= max_groups:
            PPG = d
            break
    n_groups = A * (B // PPG) if B > 0 else A
    P = A * B if B > 0 else A
    for weights in (None, "50,50", "90,10", "60,30,10", "34,33,33"):
        if weights is None:
            monkeypatch.delenv("SK_RANK_W", raising=False)
        else:
            monkeypatch.setenv("SK_RANK_W", weights)
        first, slot, ppg = np.zeros(n_groups, np.int64), np.zeros(n_groups, np.int64), np.zeros(n_groups, np.int32)
        nr = _lib.load().sk_plan_group_chunks(A, B, PPG, max_groups, G, 4, 256, n_groups, first.ctypes.data, slot.ctypes.data, ppg.ctypes.data)
        assert nr >= 1
        seen = np.zeros(P, np.int32)
        for gi in range(n_groups):
            lo, hi = int(first[gi]), int(first[gi]) + int(ppg[gi])
            assert hi <= P
            if B > 0:
                assert lo // B == (hi - 1) // B == slot[gi] // (B // PPG)     # one row a per group, filed under that row
            seen[lo:hi] += 1
        assert seen.min() == 1 and seen.max() == 1, (weights, nr)
        assert len(set(slot.tolist())) == n_groups and slot.max() == n_groups - 1


def test_launch_paths_read_no_environment_and_assume_no_cu_count():
    """SURVEY 8(b) "stateless and re-entrant": the SK_* tuning knobs are parsed ONCE, when the library is loaded (sk_abi.hip:
    parse_knobs), into an immutable struct -- no launch translation unit may call getenv; and none may hard-code MI355X's 256
    compute units (device_cu_count() asks the runtime) or keep a mutable function-local static."""
    import glob
    import re
    csrc = os.path.join(ROOT, "sigkernel_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        text = open(path).read()
        code = "\n".join(l.split("//")[0] for l in text.split("\n"))
        name = os.path.basename(path)
        if name != "sk_abi.hip":
            assert "getenv" not in code, name
        assert "256LL" not in code, name
        for m in re.finditer(r"\bstatic\s+(?!const\b|constexpr\b|inline\b|_assert)(\w+)", code):
            assert False, "%s: mutable static `%s`" % (name, m.group(0))


def test_host_layer_reads_its_route_switches_once(monkeypatch):
    """The Python layer's route switches (SK_NO_FUSED_*): parsed from the environment when the package is imported, flipped through
    `sigkernel_amd.routes` afterwards -- no call path looks at os.environ (the round-2 review counted five look-ups per call)."""
    import sigkernel_amd
    from sigkernel_amd import _routes
    for mod in ("sigkernel.py", "distributed.py", "stats.py", "static_kernels.py", "transforms.py"):
        assert "os.environ" not in open(os.path.join(ROOT, "sigkernel_amd", mod)).read(), mod
    lib_src = open(os.path.join(ROOT, "sigkernel_amd", "_lib.py")).read()
    assert "os.environ.get(\"SK_" not in lib_src
    r = _routes.Routes()
    assert not any(getattr(r, a) for a in _routes._ENV)            # (the test environment sets none of them)
    monkeypatch.setenv("SK_NO_FUSED_RBF", "1")
    assert not sigkernel_amd.routes.no_fused_rbf                   # the environment is not consulted again ...
    r.reload()
    assert r.no_fused_rbf and not r.no_fused_mb                    # ... until asked
    assert set(_routes._ENV) == set(_routes.Routes.__slots__)


VARIANTS_TABLE = os.path.join(ROOT, "profiles", "r06_variants.txt")


def _variants_table():
    """profiles/r06_variants.txt: one row per kernel instance of the build -- unit, VGPRs, scratch, launches in tools/reach_sweep.py, names."""
    rows = []
    for ln in open(VARIANTS_TABLE):
        p = ln.rstrip("\n").split("\t")
        if len(p) >= 9 and p[0] != "unit":
            rows.append(dict(unit=p[0], vgpr=int(p[1]), scratch=int(p[4]), launches=int(p[6]), instance=p[7], mangled=p[8]))
    return rows


def test_kernel_variants_stay_within_their_budget():
    """Every kernel instance of the build (tools/variants.py over the ISA csrc/Makefile keeps): none may start spilling or spill more
    than the committed table says (profiles/r06_variants.txt: VGPRs, scratch bytes, launches per instance), the instance count / library
    size stay under the round's budget -- a new route pays for its variants by removing others -- and NO INSTANCE IS UNREACHABLE: the table
    lists exactly the instances of the build, each with the launches tools/reach_sweep.py gave it (55 of 512 had none)."""
    import subprocess
    rows = _variants_table()
    assert len(rows) > 400
    dead = [r["instance"] for r in rows if r["launches"] <= 0]
    assert not dead, "kernel instances no call of tools/reach_sweep.py launches: %s" % dead
    obj = os.path.join(ROOT, "sigkernel_amd", "csrc", "obj")
    if not glob.glob(os.path.join(obj, "*gfx950.s")):
        pytest.skip("no ISA listings (the library was built elsewhere)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variants.py"), "--check", VARIANTS_TABLE], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variants.py")], capture_output=True, text=True)
    built = set(ln.split("\t")[-1] for ln in r.stdout.splitlines()[1:] if ln.strip())
    n = len(built)
    assert 0 < n <= 540, n
    listed = set(r["mangled"] for r in rows)
    assert built == listed, "instances of the build without a row in %s (run tools/reach_sweep.py on a GPU box and regenerate it): %s; rows without an instance: %s" % (
        os.path.basename(VARIANTS_TABLE), sorted(built - listed)[:5], sorted(listed - built)[:5])
    assert os.path.getsize(os.path.join(ROOT, "sigkernel_amd", "libsigkernel_amd.so")) <= 9 * (1 << 20)


def test_baseline_kernels_keep_their_waves_per_simd():
    """The kernels the BASELINE configs spend their time in, and the resident waves per SIMD their VGPR count allows (512 registers per lane
    and SIMD, allocated in eights): a compiler or source change that pushes one of them over its line -- C4's edge-keeping forward sits at
    exactly 168 = three waves -- costs a wave per SIMD silently; here it fails a test.  (The headline variant is held to two waves by its
    20 KB of LDS per wave, not by registers.)"""
    import subprocess
    obj = os.path.join(ROOT, "sigkernel_amd", "csrc", "obj")
    if not glob.glob(os.path.join(obj, "*gfx950.s")):
        pytest.skip("no ISA listings (the library was built elsewhere)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variants.py")], capture_output=True, text=True)
    vgpr = {ln.split("\t")[6]: int(ln.split("\t")[1]) for ln in r.stdout.splitlines()[1:] if ln.strip()}
    need = {"k_fwd_fused<double, 1, false, false, false, 0, 8, 4>": 2,          # C3 (headline)
            "k_fwd_fused<double, 1, false, false, false, 1, 4, 4>": 2,          # C2
            "k_fwd_fused<double, 1, false, false, true, 1, 4, 4>": 2,           # the training-sized steps' forward
            "k_adj_fused_rbf<1, 2, false, 4, 0, false>": 2,                        # ... and adjoint
            "k_fwd_fused<double, 2, false, false, true, 1, 4, 2>": 3,           # C4: forward with edges
            "k_fwd_fused<double, 2, false, true, false, 1, 4, 0>": 3,           # C4: K_YY
            "k_adj_fused_rbf<2, 1, true, 4, 0, false>": 2,                         # C4: adjoint of K_XY
            "k_adj_fused_rbf<2, 1, true, 4, 1, false>": 2,                          # C4: triangle of K_XX with second-argument sums
            "k_fwd_fused_mb<float, 2, true, 1, 16, false, 1, false>": 2}        # C5
    for name, waves in need.items():
        assert name in vgpr, name
        have = 512 // ((vgpr[name] + 7) // 8 * 8)
        assert have >= waves, "%s: %d VGPRs = %d waves per SIMD, needs %d" % (name, vgpr[name], have, waves)


@pytest.mark.gpu
def test_every_kernel_instance_is_launched_by_the_sweep():
    """The reach sweep itself, on the GPU (a minute): every route of the public API plus the batch-size- and length-gated calls, with the
    library counting its own launches per kernel instance (sk_launch_trace) -- every instance the build contains is launched at least
    once, and a sample of the swept Gram matrices and gradients matches the CPU oracle."""
    from sigkernel_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reach_sweep
    was = _lib.launch_trace(True)
    try:
        counts = reach_sweep.run(check=0.03, verbose=False)
    finally:
        _lib.launch_trace(was)
    launched = set(k.split(".kd")[0] for k, v in counts.items() if v > 0)
    missing = [r["instance"] for r in _variants_table() if r["mangled"] not in launched]
    assert not missing, missing
