#!/usr/bin/env python3
"""Achieved gradient error of the analytic adjoint (CPU oracle through the package's own autograd wiring) against every
gradient the reference fixtures hold.  The reference differentiates the static kernel by a forward difference with
h = 1e-9 (sigkernel.py:313-341, :472-500), so its gradients carry O(1e-7..1e-5) round-off noise
(tests/test_oracle.py::test_adjoint_vs_noise_free_reference_formula); the analytic adjoint is compared with a tolerance of
3x the error measured here (floor 1e-6 = north_star's bar), recorded per fixture and per gradient in grad_errors.json.

    python tests/golden/measure_grad_errors.py        # rewrites tests/golden/grad_errors.json
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

import sigkernel_amd  # noqa: E402
from sigkernel_amd import _lib  # noqa: E402
from conftest import golden, golden_gram_cases, make_kernel, rel_err  # noqa: E402
from fake_backend import OracleBackend  # noqa: E402


def main():
    _lib.set_backend(OracleBackend())
    out = {}
    for name in golden_gram_cases() + ["readme_c1"]:
        c = golden(name)
        if name == "readme_c1":
            sk = sigkernel_amd.SigKernel(sigkernel_amd.RBFKernel(sigma=0.5), 1)
        else:
            sk = sigkernel_amd.SigKernel(make_kernel(c), int(c["dyadic"]), _naive_solver=bool(c["naive"]))
        X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
        e = {}
        if "grad_w" in c:
            Xg = X.clone().requires_grad_(True)
            (sk.compute_Gram(Xg, Y) * torch.from_numpy(c["w"])).sum().backward()
            e["grad_w"] = rel_err(Xg.grad.numpy(), c["grad_w"])
        if "grad_paired" in c:
            n = c["paired"].shape[0]
            Xg = X[:n].clone().requires_grad_(True)
            (sk.compute_kernel(Xg, Y[:n]) * torch.from_numpy(c["wp"])).sum().backward()
            e["grad_paired"] = rel_err(Xg.grad.numpy(), c["grad_paired"])
        if "grad_mmd" in c:
            Xg = X.clone().requires_grad_(True)
            sk.compute_mmd(Xg, Y).backward()
            e["grad_mmd"] = rel_err(Xg.grad.numpy(), c["grad_mmd"])
        if "grad_xx_sum" in c:
            Xg = X.clone().requires_grad_(True)
            sk.compute_Gram(Xg, Xg, sym=True).sum().backward()
            e["grad_xx_sum"] = rel_err(Xg.grad.numpy(), c["grad_xx_sum"])
        if "grad_kernel_sum" in c:
            Xg = X.clone().requires_grad_(True)
            sk.compute_kernel(Xg, Y).sum().backward()
            e["grad_kernel_sum"] = rel_err(Xg.grad.numpy(), c["grad_kernel_sum"])
        out[name] = e
        print(name, {k: "%.2e" % v for k, v in e.items()})
    json.dump(out, open(os.path.join(HERE, "grad_errors.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
