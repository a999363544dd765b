"""mit_semseg.config (the yacs-free option tree of the drivers) against the reference's shipped YAML files
(tests/golden/config_golden.json, made by tests/golden/make_config_golden.py) and the override semantics train.py / eval.py /
test.py rely on (train.py:225-262)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_golden.json')))


def _flat(d, prefix=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + '.'))
        else:
            out[prefix + k] = list(v) if isinstance(v, tuple) else v
    return out


@pytest.mark.parametrize('name', sorted(GOLD))
def test_presets_equal_reference_yaml(name, tmp_path):
    from mit_semseg.config import cfg, preset, load, PRESETS
    assert set(PRESETS) == set(GOLD)
    want = _flat(cfg.to_dict())
    want.update(_flat(GOLD[name]))              # defaults <- the reference's file
    assert _flat(preset(name).to_dict()) == want
    # through a YAML file (what `--cfg FILE` does): dump -> load must be the identity
    p = tmp_path / (name + '.yaml')
    p.write_text(preset(name).dump())
    assert _flat(load(str(p)).to_dict()) == want
    assert _flat(load('preset:' + name).to_dict()) == want


def test_yaml_string_scalars_and_overrides(tmp_path):
    from mit_semseg.config import load
    p = tmp_path / 'c.yaml'
    p.write_text('DATASET:\n  imgSizes: (300, 375)\n  imgMaxSize: 800\nTRAIN:\n  weight_decay: 1e-5\n  fix_bn: True\nDIR: "ckpt/x"\n')
    c = load(str(p), ['TRAIN.lr_encoder', '0.01', 'MODEL.fc_dim', '512', 'DATASET.imgSizes', '(450,)', 'VAL.visualize', 'True'])
    assert c.DATASET.imgSizes == (450,) and c.DATASET.imgMaxSize == 800
    assert c.TRAIN.weight_decay == 1e-5 and c.TRAIN.fix_bn is True and c.TRAIN.lr_encoder == 0.01
    assert c.MODEL.fc_dim == 512 and c.VAL.visualize is True and c.DIR == 'ckpt/x'
    c.TRAIN.max_iters = c.TRAIN.epoch_iters * c.TRAIN.num_epoch          # drivers add keys (train.py:255-259)
    assert c.TRAIN.max_iters == 100000
    assert 'max_iters: 100000' in str(c)
    with pytest.raises(KeyError):
        load(str(p), ['TRAIN.no_such_key', '1'])
    with pytest.raises(ValueError):
        load(str(p), ['TRAIN.num_epoch', 'many'])
    c.freeze()
    with pytest.raises(AttributeError):
        c.TRAIN.num_epoch = 3
    d = c.clone()
    d.defrost()
    d.TRAIN.num_epoch = 3
    assert c.TRAIN.num_epoch == 20 and d.TRAIN.num_epoch == 3


def test_import_surface_of_the_reference_drivers():
    """what train.py:12-16, eval.py:12-19, test.py:10-19 import"""
    from mit_semseg.config import cfg                                                            # noqa: F401
    from mit_semseg.dataset import TrainDataset, ValDataset, TestDataset                           # noqa: F401
    from mit_semseg.models import ModelBuilder, SegmentationModule                                 # noqa: F401
    from mit_semseg.utils import (AverageMeter, parse_devices, setup_logger, colorEncode, accuracy,   # noqa: F401
                                  intersectionAndUnion, find_recursive)
    from mit_semseg.lib.nn import (UserScatteredDataParallel, user_scattered_collate, patch_replication_callback,  # noqa: F401
                                   async_copy_to)
    from mit_semseg.lib.utils import as_numpy                                                      # noqa: F401
    assert parse_devices('0-3') == ['gpu0', 'gpu1', 'gpu2', 'gpu3']
    assert parse_devices('gpu1,3, 5-6') == ['gpu1', 'gpu3', 'gpu5', 'gpu6']
    assert parse_devices('2-0,1') == ['gpu0', 'gpu1', 'gpu2']
    with pytest.raises(Exception):
        parse_devices('tpu0')
