#!/usr/bin/env python3
"""Round-off noise of every gradient the reference fixtures hold, measured against the reference's OWN formula evaluated in
long double (tests/ld_reference.py) -- not against any implementation: this script imports neither sigkernel_amd nor the
reference.  The reference differentiates the static kernel by a forward difference with h = 1e-9 in double precision
(sigkernel.py:313-341, :472-500); what separates a fixture from the long-double evaluation of the same formula is that
difference's cancellation noise, and it is the only thing a comparison tolerance above north_star's 1e-6 may rest on
(tests/conftest.py::grad_tol = max(1e-6, 1.25 x the noise recorded here)).

    python tests/golden/measure_grad_noise.py        # rewrites tests/golden/grad_errors.json
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

from ld_reference import reference_gradient_ld, rel_err_ld  # noqa: E402

KEYS = ("grad_w", "grad_paired", "grad_mmd", "grad_xx_sum", "grad_kernel_sum")


def main():
    out = {}
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "gram_*.npz"))) + ["readme_c1"]
    for name in names:
        c = dict(np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False))
        kw = dict(kernel="rbf", param=float(c["sigma"])) if name == "readme_c1" else {}
        out[name] = {k: {"reference_noise": rel_err_ld(c[k], reference_gradient_ld(c, k, **kw))} for k in KEYS if k in c}
        print(name, {k: "%.2e" % v["reference_noise"] for k, v in out[name].items()})
    out["_provenance"] = ("max-norm relative distance of each fixture gradient from the reference's formula in long double; "
                          "tests/golden/measure_grad_noise.py (imports no implementation)")
    json.dump(out, open(os.path.join(HERE, "grad_errors.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
