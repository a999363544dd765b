"""Pretrained-checkpoint lookup behind `weights=''` (the contract of the reference's models/utils.py:10-18: the file
named by the URL's last component, kept under ./pretrained, loaded on the CPU).

This build runs without a network, so the lookup is cache-first and says what to do on a miss: a checkpoint that is
already under `model_dir` (or under $SEMSEG_PRETRAINED_DIR) is loaded; only otherwise is a fetch attempted, and a
failed fetch names the file the caller has to put there instead of leaving a half-written one behind."""
import os
import pathlib
import urllib.error
import urllib.request

import torch


def _candidates(name, model_dir):
    yield pathlib.Path(model_dir) / name
    extra = os.environ.get('SEMSEG_PRETRAINED_DIR')
    if extra:
        yield pathlib.Path(extra) / name


def load_url(url, model_dir='./pretrained', map_location=None):
    name = url.rsplit('/', 1)[-1]
    for path in _candidates(name, model_dir):
        if path.is_file():
            return torch.load(str(path), map_location=map_location)
    target = pathlib.Path(model_dir) / name
    target.parent.mkdir(parents=True, exist_ok=True)
    partial = target.with_suffix(target.suffix + '.part')
    try:
        with urllib.request.urlopen(url, timeout=30) as src, open(partial, 'wb') as dst:
            while True:
                chunk = src.read(1 << 20)
                if not chunk:
                    break
                dst.write(chunk)
        partial.replace(target)
    except (urllib.error.URLError, OSError) as e:
        if partial.exists():
            partial.unlink()
        raise RuntimeError(
            'pretrained weights %s are not under %s and could not be fetched from %s (%s); place the file there '
            'or pass weights=<path to a state_dict>' % (name, model_dir, url, e)) from e
    return torch.load(str(target), map_location=map_location)
