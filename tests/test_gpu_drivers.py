"""The command-line drivers (semantic-segmentation-pytorch_amd/{train,eval_multipro,test}.py = train.py:208-273, eval.py:150-193,
test.py:130-200 of the reference) end to end on a tiny synthetic dataset: two epochs with epoch checkpoints, resume from the
first, validation of the second checkpoint (mIoU / accuracy lines), inference on image files."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')


def _dataset(root):
    from PIL import Image
    rng = np.random.RandomState(7)
    os.makedirs(os.path.join(root, 'images'), exist_ok=True)
    os.makedirs(os.path.join(root, 'annotations'), exist_ok=True)
    recs = []
    for i, (h, w) in enumerate([(96, 128), (96, 128), (128, 96), (128, 96), (100, 140), (90, 120)]):
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        seg = rng.randint(0, 151, size=(h // 8 + 1, w // 8 + 1)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w]
        Image.fromarray(img).save(os.path.join(root, 'images', 'im%d.jpg' % i), quality=95)
        Image.fromarray(seg, mode='L').save(os.path.join(root, 'annotations', 'im%d.png' % i))
        recs.append({'fpath_img': 'images/im%d.jpg' % i, 'fpath_segm': 'annotations/im%d.png' % i, 'width': w, 'height': h})
    for name, rs in (('train.odgt', recs), ('val.odgt', recs[:3])):
        with open(os.path.join(root, name), 'w') as f:
            f.write('\n'.join(json.dumps(r) for r in rs) + '\n')
    return recs


def _run(script, args, env):
    r = subprocess.run([sys.executable, os.path.join(PKG, script)] + args, cwd=PKG, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (script, r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_train_checkpoint_resume_eval_test(tmp_path):
    data = str(tmp_path / 'data')
    _dataset(data)
    ckpt = str(tmp_path / 'ckpt')
    # MODEL.weights_encoder '' means "download the ImageNet weights" (models.py:65), as in the reference: start from a seeded
    # backbone file instead (the oracle recipe of SURVEY 8c)
    sys.path.insert(0, PKG)
    from mit_semseg.models import resnet
    from mit_semseg.models.models import ResnetDilated
    torch.manual_seed(1)
    enc0 = str(tmp_path / 'enc0.pth')
    torch.save(ResnetDilated(resnet.resnet18(pretrained=False), dilate_scale=8).state_dict(), enc0)
    env = dict(os.environ, SEMSEG_TUNE='0', PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    common = ['--cfg', 'preset:ade20k-resnet18dilated-ppm_deepsup']
    opts = ['DIR', ckpt, 'DATASET.root_dataset', data, 'DATASET.list_train', os.path.join(data, 'train.odgt'),
            'DATASET.list_val', os.path.join(data, 'val.odgt'), 'DATASET.imgSizes', '(64, 96)', 'DATASET.imgMaxSize', '160',
            'TRAIN.epoch_iters', '3', 'TRAIN.num_epoch', '2', 'TRAIN.disp_iter', '1']
    out = _run('train.py', common + ['--gpus', '0'] + opts + ['MODEL.weights_encoder', enc0], env)
    assert 'Training Done!' in out and out.count('Epoch: [') == 6, out[-2000:]
    for e in (1, 2):
        for part in ('encoder', 'decoder', 'history'):
            assert os.path.exists(os.path.join(ckpt, '%s_epoch_%d.pth' % (part, e))), (part, e)
    assert os.path.exists(os.path.join(ckpt, 'config.yaml'))
    hist = torch.load(os.path.join(ckpt, 'history_epoch_2.pth'), weights_only=False)
    assert len(hist['train']['loss']) == 6 and all(np.isfinite(hist['train']['loss']))
    enc1 = torch.load(os.path.join(ckpt, 'encoder_epoch_1.pth'), weights_only=False)
    enc2 = torch.load(os.path.join(ckpt, 'encoder_epoch_2.pth'), weights_only=False)
    assert set(enc1) == set(enc2) and 'layer4.1.conv2.weight' in enc1 and 'bn1._running_iter' in enc1
    assert not torch.equal(enc1['conv1.weight'], enc2['conv1.weight'])
    assert int(enc2['bn1.num_batches_tracked']) == 6 and float(enc2['bn1._running_iter']) == 1000.0     # 1 / momentum
    # resume: start_epoch 1 loads epoch 1's files and trains epoch 2 again (train.py:240-247)
    out = _run('train.py', common + ['--gpus', '0'] + opts + ['TRAIN.start_epoch', '1'], env)
    assert out.count('Loading weights for net_encoder') == 1 and out.count('Epoch: [2]') == 3 and 'Epoch: [1]' not in out
    # validation of epoch 2 (eval.py): summary lines, per-class IoU lines
    out = _run('eval_multipro.py', common + ['--gpu', '0'] + opts + ['VAL.checkpoint', 'epoch_2.pth', 'VAL.visualize', 'True'], env)
    assert 'Evaluation Done!' in out and 'Mean IoU:' in out and out.count('class [') == 150, out[-1500:]
    assert len([f for f in os.listdir(os.path.join(ckpt, 'result')) if f.endswith('.png')]) == 3
    # inference on files (test.py)
    res = str(tmp_path / 'res')
    out = _run('test.py', ['--imgs', os.path.join(data, 'images'), '--gpu', '0'] + common + opts +
               ['TEST.checkpoint', 'epoch_2.pth', 'TEST.result', res], env)
    assert 'Inference done!' in out
    assert len([f for f in os.listdir(res) if f.endswith('.png')]) == 6
