"""Path transforms applied before the kernel in every example of the reference (transformers.py:12-80), as batched
tensor ops that run on the device the paths live on (the reference maps numpy lists one path at a time on the CPU).

All take and return tensors of shape (batch, length, dim).
"""
import torch

__all__ = ["add_time", "lead_lag", "transform"]


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
