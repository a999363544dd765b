"""The seven shipped configurations of the reference (config/ade20k-*.yaml) as overrides of the defaults, so that the drivers
work out of the box where the reference's YAML files are not at hand:

    python -m mit_semseg.config --list
    python -m mit_semseg.config --write ade20k-resnet50dilated-ppm_deepsup > my.yaml
    python train.py --cfg preset:ade20k-hrnetv2

Only what differs from config/defaults.py is listed (model strings, fc_dim, label stride / padding, epochs, checkpoints)."""
from .defaults import _C


def _p(enc, dec, fc_dim, rate=8, pad=8, epochs=20, ckpt=None, name=None, batch=2):
    ck = ckpt or 'epoch_%d.pth' % epochs
    return name or 'ade20k-%s-%s' % (enc, dec), {
                  'DIR': 'ckpt/ade20k-%s-%s' % (enc, dec),
                  'DATASET': {'segm_downsampling_rate': rate, 'padding_constant': pad},
                  'MODEL': {'arch_encoder': enc, 'arch_decoder': dec, 'fc_dim': fc_dim},
                  'TRAIN': {'num_epoch': epochs, 'batch_size_per_gpu': batch},
                  'VAL': {'checkpoint': ck}, 'TEST': {'checkpoint': ck}}


PRESETS = dict([
    _p('mobilenetv2dilated', 'c1_deepsup', 320, batch=3),
    _p('resnet18dilated', 'ppm_deepsup', 512),
    _p('resnet50dilated', 'ppm_deepsup', 2048),
    _p('resnet101dilated', 'ppm_deepsup', 2048, epochs=25),
    _p('resnet50', 'upernet', 2048, rate=4, pad=32, epochs=30),
    _p('resnet101', 'upernet', 2048, rate=4, pad=32, epochs=40, ckpt='epoch_50.pth'),      # the shipped file names epoch 50
    _p('hrnetv2', 'c1', 720, rate=4, pad=32, epochs=30, name='ade20k-hrnetv2'),
])


def preset(name):
    cfg = _C.clone()
    cfg.merge_from_other_cfg(PRESETS[name])
    return cfg
