"""API shims for reference lib/nn/parallel/data_parallel.py:13-112.

`UserScatteredDataParallel(module, device_ids)` in the reference scatters element i of the input
list to GPU i inside ONE process.  Here every rank is its own process (torchrun, RCCL), so the
wrapper takes this rank's element of the list (or the single dict) and runs the resident module;
gradients are all-reduced by mit_semseg.parallel.  `user_scattered_collate` and `async_copy_to`
keep their reference semantics.
"""
import collections.abc

import torch

from ....parallel import NativeDataParallel

__all__ = ['UserScatteredDataParallel', 'user_scattered_collate', 'async_copy_to']


def async_copy_to(obj, dev, main_stream=None):
    """Reference data_parallel.py:13-24: recursive non-blocking H2D copy."""
    if torch.is_tensor(obj):
        v = obj.cuda(dev, non_blocking=True)
        if main_stream is not None:
            v.record_stream(main_stream)
        return v
    if isinstance(obj, collections.abc.Mapping):
        return {k: async_copy_to(o, dev, main_stream) for k, o in obj.items()}
    if isinstance(obj, collections.abc.Sequence) and not isinstance(obj, str):
        return [async_copy_to(o, dev, main_stream) for o in obj]
    return obj


def user_scattered_collate(batch):
    """Reference data_parallel.py:65-66: identity (the dataset already yields per-GPU dicts)."""
    return batch


class UserScatteredDataParallel(NativeDataParallel):
    """Reference data_parallel.py:48-62."""
