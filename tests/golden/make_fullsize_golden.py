"""Full-size forward fixtures of the UNMODIFIED reference for BASELINE configs[1] and configs[2] (2 x 512 x 512, training-mode
forward: batch-statistics BN, replayed Dropout2d masks), so that the 512 x 512 parity test compares the HIP path with the reference
itself and not only with the oracle port (round-3 review).  Same recipe as tests/test_gpu_models.py::test_full_size_vs_oracle:
synthetic weights seed 3 / 4, dropout masks seed 3 / 4, batch seed 307.

    PYTHONPATH=/root/reference python tests/golden/make_fullsize_golden.py        (build container only)

Stored per case (tests/golden/fullsize/<case>.pt): loss, acc, the arg-max label map and the top-1 / top-2 margin of EVERY pixel,
and the 150 log-probabilities of a seeded pixel sample (all 8192 pixels for configs[1]; 8192 of the 32768 of configs[2]) --
about 5 MB each."""
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                      # noqa: E402  (puts /root/reference first on sys.path, asserts the import is the reference's)
from oracle import semseg_oracle as O        # noqa: E402

CASES = {
    'cfg1_r50d_ppmds_512': ('resnet50dilated', 'ppm_deepsup', 2048, 0.4, 8, 512, 512),
    'cfg2_r50_upernet_512': ('resnet50', 'upernet', 2048, None, 4, 512, 512),
}
SAMPLE = 8192


def pixel_sample(npix, k=SAMPLE):
    """seeded pixel sample shared with the test (tests/test_gpu_models.py rebuilds it): all pixels when there are <= k"""
    if npix <= k:
        return torch.arange(npix)
    return torch.randperm(npix, generator=torch.Generator().manual_seed(npix))[:k].sort()[0]


def run(case):
    arch_enc, arch_dec, fc_dim, dss, rate, H, W = CASES[case]
    torch.manual_seed(304)
    enc, dec = MG.build_reference(arch_enc, arch_dec, fc_dim)
    enc.load_state_dict(O.synth_state_dict(MG.manifest_of(enc), 3))
    dec.load_state_dict(O.synth_state_dict(MG.manifest_of(dec), 4))
    drop = {}
    if 'ppm' in arch_dec:
        drop['main'] = O.synth_dropout_mask(2, 512, seed=3)
        dec.conv_last[3] = MG.ReplayDropout(drop['main'])
    if arch_dec == 'ppm_deepsup':
        drop['deepsup'] = O.synth_dropout_mask(2, fc_dim // 4, seed=4)
        dec.dropout_deepsup = MG.ReplayDropout(drop['deepsup'])
    sm = MG.SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), dss)
    sm.train()
    img, lab = O.synth_batch(2, H, W, rate, seed=307)
    cap = {}
    hk = dec.register_forward_hook(lambda m, i, o: cap.__setitem__('out', o))
    with torch.no_grad():
        loss, acc = sm({'img_data': img, 'seg_label': lab})
    hk.remove()
    o = cap['out']
    pred = (o[0] if isinstance(o, tuple) else o).detach()
    n, c, h, w = pred.shape
    rows = pred.permute(0, 2, 3, 1).reshape(n * h * w, c)
    top2 = rows.topk(2, dim=1)[0]
    idx = pixel_sample(n * h * w)
    return {'meta': dict(case=case, arch_encoder=arch_enc, arch_decoder=arch_dec, fc_dim=fc_dim, deep_sup_scale=dss, seg_rate=rate,
                         n=2, h=H, w=W, seed_weights=(3, 4), seed_dropout=(3, 4), seed_batch=307, torch=torch.__version__,
                         pred_shape=list(pred.shape)),
            'loss': loss.detach().clone(), 'acc': acc.detach().clone(),
            'argmax': rows.argmax(1).to(torch.int16).reshape(n, h, w).clone(),
            'margin': (top2[:, 0] - top2[:, 1]).reshape(n, h, w).clone(),
            'pixels': idx.clone(), 'logp': rows[idx].clone()}


def main():
    torch.set_num_threads(os.cpu_count())
    out_dir = os.path.join(HERE, 'fullsize')
    os.makedirs(out_dir, exist_ok=True)
    for case in (sys.argv[1:] or sorted(CASES)):
        r = run(case)
        path = os.path.join(out_dir, case + '.pt')
        torch.save(r, path)
        print('%-24s %7.1f KB loss=%.6f acc=%.6f min margin %.2e' % (case, os.path.getsize(path) / 1024, r['loss'].item(),
                                                                    r['acc'].item(), r['margin'].min().item()))


if __name__ == '__main__':
    main()
