"""Default configuration of the drivers: the option tree of reference mit_semseg/config/defaults.py:1-97 (same section and key
names, same default values -- the reference's YAML files and `opts` overrides apply unchanged), held in this build's own
CfgNode (config/node.py)."""
from .node import CfgNode as CN

_C = CN({
    'DIR': 'ckpt/ade20k-resnet50dilated-ppm_deepsup',
    'DATASET': {
        'root_dataset': './data/', 'list_train': './data/training.odgt', 'list_val': './data/validation.odgt',
        'num_class': 150,
        'imgSizes': (300, 375, 450, 525, 600),      # multi-scale short sides for training, the one test scale at evaluation
        'imgMaxSize': 1000,                         # bound on the long side
        'padding_constant': 8,                      # batch H, W are padded to multiples of this
        'segm_downsampling_rate': 8,                # label map stride of the network
        'random_flip': True,
    },
    'MODEL': {'arch_encoder': 'resnet50dilated', 'arch_decoder': 'ppm_deepsup', 'weights_encoder': '', 'weights_decoder': '',
              'fc_dim': 2048},
    'TRAIN': {
        'batch_size_per_gpu': 2, 'num_epoch': 20, 'start_epoch': 0, 'epoch_iters': 5000, 'optim': 'SGD',
        'lr_encoder': 0.02, 'lr_decoder': 0.02, 'lr_pow': 0.9, 'beta1': 0.9, 'weight_decay': 1e-4, 'deep_sup_scale': 0.4,
        'fix_bn': False, 'workers': 16, 'disp_iter': 20, 'seed': 304,
    },
    'VAL': {'batch_size': 1, 'visualize': False, 'checkpoint': 'epoch_20.pth'},
    'TEST': {'batch_size': 1, 'checkpoint': 'epoch_20.pth', 'result': './'},
})
