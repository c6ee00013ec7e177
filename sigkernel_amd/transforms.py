"""Path transforms applied before the kernel in every example of the reference (transformers.py:12-80), as batched
tensor ops that run on the device the paths live on (the reference maps numpy lists one path at a time on the CPU).

The functions take and return tensors of shape (batch, length, dim).  `AddTime` / `LeadLag` are the reference's sklearn-style
transformer classes (transformers.py:30-44, :57-80) over the same code: `fit` / `transform` / `fit_transform` / `transform_instance`,
constructor arguments as there; they accept what the reference's accept (a list of per-path arrays, ragged lengths included, or one
array) and, additionally, a (batch, length, dim) tensor, which stays a tensor on its device.
"""
import numpy as np
import torch

__all__ = ["add_time", "lead_lag", "transform", "AddTime", "LeadLag"]

try:    # the reference's classes are sklearn estimators (pipelines, get_params); without sklearn they are plain objects
    from sklearn.base import BaseEstimator, TransformerMixin
except Exception:      # noqa: BLE001
    class BaseEstimator:
        pass

    class TransformerMixin:
        def fit_transform(self, X, y=None, **fit_params):
            return self.fit(X, y, **fit_params).transform(X)


def add_time(paths, init_time=0.):
    """Prepend a time channel running over [init_time, init_time + 1] (AddTime.transform_instance, transformers.py:39-41;
    like the reference, the ``total_time`` attribute is not used)."""
    B, L, _ = paths.shape
    t = torch.linspace(init_time, init_time + 1, L, dtype=paths.dtype, device=paths.device)
    return torch.cat([t.reshape(1, L, 1).expand(B, L, 1), paths], dim=2)


def lead_lag(paths):
    """Lead-lag transform (LeadLag.transform_instance, transformers.py:63-77): length 2L-1, channels [lag, lead];
    lag = x0,x0,x1,x1,...,x_{L-1};  lead = x0,x1,x1,x2,...,x_{L-1},x_{L-1}."""
    B, L, D = paths.shape
    rep = torch.repeat_interleave(paths, 2, dim=1)          # x0,x0,x1,x1,...,x_{L-1},x_{L-1}
    lag, lead = rep[:, :-1], rep[:, 1:]
    return torch.cat([lag, lead], dim=2)


def transform(paths, at=False, ll=False, scale=1.):
    """scale, then lead-lag, then add-time -- the reference's ``transform`` (transformers.py:12-18)."""
    paths = scale * paths
    if ll:
        paths = lead_lag(paths)
    if at:
        paths = add_time(paths)
    return paths


class _PathTransformer(BaseEstimator, TransformerMixin):
    """A (batch, length, dim) tensor goes through the batched op in one piece; anything else is the reference's interface: an
    iterable of per-path arrays (length, dim) -- possibly of different lengths -- mapped one by one to a list of numpy arrays."""

    def fit(self, X, y=None):
        return self

    def _op(self, paths):
        raise NotImplementedError

    def transform_instance(self, X):
        x = torch.as_tensor(np.asarray(X, dtype=np.float64))
        if x.dim() == 1:
            x = x[:, None]
        return self._op(x[None])[0].numpy()

    def transform(self, X, y=None):
        if isinstance(X, torch.Tensor) and X.dim() == 3:
            return self._op(X)
        return [self.transform_instance(x) for x in X]


class AddTime(_PathTransformer):
    """The reference's ``AddTime`` (transformers.py:30-44): a leading time channel over [init_time, init_time + 1]
    (``total_time`` is stored and, as in the reference, not used)."""

    def __init__(self, init_time=0., total_time=1.):
        self.init_time = init_time
        self.total_time = total_time

    def _op(self, paths):
        return add_time(paths, self.init_time)


class LeadLag(_PathTransformer):
    """The reference's ``LeadLag`` (transformers.py:57-80): length 2L-1, channels [lag, lead]."""

    def __init__(self):
        pass

    def _op(self, paths):
        return lead_lag(paths)
