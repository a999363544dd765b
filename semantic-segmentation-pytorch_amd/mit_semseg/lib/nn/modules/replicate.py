"""API shims for reference lib/nn/modules/replicate.py:27-94.

The reference patches nn.DataParallel's per-iteration replicate() so every SyncBN copy finds its
master pipe.  With one process per GPU there is nothing to replicate: parameters are resident on
each rank and BN statistics travel over RCCL.  The names are kept so `train.py:185-189`-style code
runs unchanged; they delegate to mit_semseg.parallel.
"""
from ....parallel import NativeDataParallel

__all__ = ['DataParallelWithCallback', 'patch_replication_callback', 'CallbackContext',
           'execute_replication_callbacks']


class CallbackContext(object):
    pass


def execute_replication_callbacks(modules):
    """No-op: there are no thread replicas (reference replicate.py:27-47)."""
    return None


class DataParallelWithCallback(NativeDataParallel):
    """Reference replicate.py:50-67.  device_ids beyond this process' own GPU are ignored."""


def patch_replication_callback(data_parallel):
    """Reference replicate.py:70-94: monkey-patches replicate(); nothing to patch here."""
    return None
