"""Build the REAL reference solver (container-only validation aid).

Cythonises /root/reference/sigkernel/cython_backend.pyx *where it lies* (the
source is never copied into this repository) and compiles it with plain
``gcc -O2`` -- mirroring /root/reference/setup.py:44-51 -- into
``oracle/_ref/cython_backend*.so``.  Also drops a stub ``numba`` package there,
because the reference imports numba unconditionally (sigkernel.py:4,
cuda_backend.py:2, static_kernels.py:4) and numba is not installed.

The reference is Python: per the travel rule nothing under oracle/_ref/ goes to
the GPU box (.gpurunignore) or into history (.gitignore).  It is used only by
tests/golden/make_golden.py and tests/test_oracle_vs_reference.py (skipped when
/root/reference is absent).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIGKERNEL_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def build(verbose=True):
    pyx = os.path.join(REF, "sigkernel", "cython_backend.pyx")
    if not os.path.exists(pyx):
        raise FileNotFoundError(pyx)
    os.makedirs(os.path.join(OUT, "numba"), exist_ok=True)
    c_file = os.path.join(OUT, "cython_backend.c")
    so = os.path.join(OUT, "cython_backend" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(pyx):
        subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_file])
        import numpy
        inc = [sysconfig.get_paths()["include"], numpy.get_include()]
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fwrapv", "-fno-strict-aliasing", "-w"]
        cmd += ["-I" + i for i in inc] + [c_file, "-o", so]
        subprocess.check_call(cmd)
    with open(os.path.join(OUT, "numba", "__init__.py"), "w") as f:
        f.write(
            "# stub: numba is absent in this image; the CPU path of the reference never calls these\n"
            "class _Cuda:\n"
            "    @staticmethod\n"
            "    def jit(f=None, **kw):\n"
            "        return f if f is not None else (lambda g: g)\n"
            "    @staticmethod\n"
            "    def as_cuda_array(x):\n"
            "        raise RuntimeError('no numba.cuda here')\n"
            "cuda = _Cuda()\n")
    if verbose:
        print("built", so)
    return OUT


def import_reference():
    """Return the imported reference package ``sigkernel`` (container only)."""
    out = build(verbose=False)
    for p in (REF, out):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sigkernel  # noqa: the reference package
    return sigkernel


if __name__ == "__main__":
    build()
