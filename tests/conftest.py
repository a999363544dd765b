import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need an MI355X and the built native library: skip them (instead of erroring) on a CPU-only box, so
    a plain `pytest tests` is green there too."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device visible (gpu-marked tests run on the MI355X box)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """the parity lines of the run (max|dlogp|, argmax flips, band ratios per case) in the log, with or without -s"""
    try:
        from tests.util import PARITY_LINES
    except Exception:
        return
    if PARITY_LINES:
        terminalreporter.section('parity lines')
        for ln in PARITY_LINES:
            terminalreporter.write_line(ln)
