#!/usr/bin/env python3
"""Time-series classification with a precomputed signature-kernel Gram matrix -- the pipeline of the reference's
examples/time_series_classification.py:94 and :189-202 (transform -> compute_Gram(sym=True) -> sklearn SVC with
kernel='precomputed', hyper-parameters by cross-validated grid search), on synthetic two-class paths: the UCR/UEA data
sets the reference downloads through tslearn are not available offline.

    python examples/time_series_classification.py [--n-train 120] [--n-test 80] [--length 60]

Class 0: Brownian paths with a slow sinusoidal drift; class 1: the same noise with the drift's frequency doubled.
Runs on an MI355X (the Gram matrices come from the HIP kernels; there is no CPU fallback).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sigkernel_amd as sigkernel  # noqa: E402


def make_dataset(n, length, seed, noise=0.35):
    """(n, length, 1) float64 paths and (n,) integer labels, two balanced classes."""
    rng = np.random.default_rng(seed)
    y = np.arange(n) % 2
    t = np.linspace(0.0, 1.0, length)
    phase = rng.uniform(0, 2 * np.pi, size=(n, 1))
    drift = np.sin(2 * np.pi * (1 + y)[:, None] * t[None, :] + phase)
    walk = np.cumsum(rng.normal(scale=noise / np.sqrt(length), size=(n, length)), axis=1)
    x = (drift + walk)[:, :, None]
    return x, y


def fit_signature_svc(x_train, y_train, device, sigmas=(0.25, 0.5, 1.0), at=True, ll=False, scale=0.1, dyadic_order=0,
                      cv=5, dtype=torch.float64):
    """Grid search over the RBF sigma of the static kernel and the SVC's C, as the reference does
    (examples/time_series_classification.py:150-202).  Returns (best cv score, sigma, fitted GridSearchCV, train tensor)."""
    from sklearn.model_selection import GridSearchCV
    from sklearn.svm import SVC
    x_train = x_train / np.abs(x_train).max()                                    # :88
    xt = sigkernel.transform(torch.tensor(x_train, dtype=dtype, device=device), at=at, ll=ll, scale=scale)   # :94
    best = (-1.0, None, None)
    for sigma in sigmas:
        signature_kernel = sigkernel.SigKernel(sigkernel.RBFKernel(sigma=sigma), dyadic_order=dyadic_order)     # :186-189
        G_train = signature_kernel.compute_Gram(xt, xt, sym=True).cpu().numpy()                                # :192
        svc = SVC(kernel="precomputed", decision_function_shape="ovo")                                         # :195
        model = GridSearchCV(estimator=svc, param_grid={"C": np.logspace(0, 4, 5)}, cv=cv, n_jobs=1)           # :196
        model.fit(G_train, y_train)                                                                            # :197
        if model.best_score_ > best[0]:
            best = (float(model.best_score_), sigma, model)
    return best + (xt,)


def predict(model, sigma, xt_train, x_test, x_train_max, device, at=True, ll=False, scale=0.1, dyadic_order=0,
            dtype=torch.float64):
    """Test-vs-train Gram matrix and the SVC's predictions (examples/time_series_classification.py:262-281)."""
    xs = sigkernel.transform(torch.tensor(x_test / x_train_max, dtype=dtype, device=device), at=at, ll=ll, scale=scale)
    signature_kernel = sigkernel.SigKernel(sigkernel.RBFKernel(sigma=sigma), dyadic_order=dyadic_order)
    G_test = signature_kernel.compute_Gram(xs, xt_train, sym=False).cpu().numpy()
    return model.predict(G_test)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-train", type=int, default=120)
    ap.add_argument("--n-test", type=int, default=80)
    ap.add_argument("--length", type=int, default=60)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("this example needs an MI355X (sigkernel_amd has no CPU path)")
    device = torch.device("cuda", 0)
    x_train, y_train = make_dataset(args.n_train, args.length, seed=0)
    x_test, y_test = make_dataset(args.n_test, args.length, seed=1)
    score, sigma, model, xt = fit_signature_svc(x_train, y_train, device)
    pred = predict(model, sigma, xt, x_test, np.abs(x_train).max(), device)
    acc = float(np.mean(pred == y_test))
    print("signature PDE kernel + SVC: cv accuracy %.3f (sigma %.2f, C %g), test accuracy %.3f"
          % (score, sigma, model.best_params_["C"], acc))
    return acc


if __name__ == "__main__":
    main()
