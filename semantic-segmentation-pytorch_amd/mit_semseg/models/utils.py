"""Checkpoint fetch helper with the reference's contract (models/utils.py:10-18): download into
./pretrained and torch.load on CPU.  There is no network in the build/bench environment, so this is
only reached when a caller passes weights='' (pretrained=True)."""
import os
import sys

import torch

try:
    from urllib import urlretrieve
except ImportError:
    from urllib.request import urlretrieve


def load_url(url, model_dir='./pretrained', map_location=None):
    if not os.path.exists(model_dir):
        os.makedirs(model_dir)
    filename = url.split('/')[-1]
    cached_file = os.path.join(model_dir, filename)
    if not os.path.exists(cached_file):
        sys.stderr.write('Downloading: "{}" to {}\n'.format(url, cached_file))
        urlretrieve(url, cached_file)
    return torch.load(cached_file, map_location=map_location)
