"""Build libsigkernel_amd.so (hipcc, gfx950) in-tree.  `python -m sigkernel_amd.build`."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsigkernel_amd.so")


def build(force=False, verbose=False):
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise RuntimeError("building libsigkernel_amd.so failed")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
