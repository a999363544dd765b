"""`mit_semseg.lib.utils` of the reference (lib/utils/__init__.py, th.py:8-41): container-recursive tensor helpers its
evaluation drivers import (`as_numpy`, eval.py:14,75).  The vendored DataLoader copy under lib/utils/data of the reference
(a 2018 fork of torch.utils.data) has no counterpart: torch's own loader serves the drivers here."""
import collections.abc

import numpy as np
import torch

__all__ = ['as_variable', 'as_numpy', 'mark_volatile']


def _map(obj, fn):
    if isinstance(obj, collections.abc.Mapping):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, collections.abc.Sequence) and not isinstance(obj, (str, bytes)):
        return [_map(v, fn) for v in obj]
    return fn(obj)


def as_variable(obj):
    """th.py:8-15: tensors stay tensors (Variable and Tensor are one type since torch 0.4)"""
    return _map(obj, lambda v: v)


def as_numpy(obj):
    """th.py:18-27: every tensor in a (nested) container as a host numpy array"""
    return _map(obj, lambda v: v.detach().cpu().numpy() if torch.is_tensor(v) else np.array(v))


def mark_volatile(obj):
    """th.py:30-41: `volatile` is gone from torch; gradients are switched off with torch.no_grad() by the drivers"""
    return _map(obj, lambda v: v.detach() if torch.is_tensor(v) else v)
