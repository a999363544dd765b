"""Full-size fixtures of the UNMODIFIED reference for BASELINE configs[1..4] (bs 2; 512 x 512, and the two variable-size shapes
the GPU test uses for configs[3]): ONE TRAINING STEP -- training-mode forward (batch-statistics BN, replayed Dropout2d masks),
`loss.backward()`, the two SGD updates (train.py:34-48) -- so that the full-size parity test compares the HIP path, forward AND
every gradient tensor, with the reference itself and not only with the oracle port (round-3 / round-4 reviews).
Same recipe as tests/test_gpu_models.py::test_full_size_vs_oracle: synthetic weights seed 3 / 4, dropout masks seed 3 / 4,
batch seed 307 -- which is make_golden.run_case(seed=3), so the six executions behind every anchor record (float64 + the five
fp32 executions of make_golden.VARIANTS) are exactly those of the small goldens.

    PYTHONPATH=/root/reference python tests/golden/make_fullsize_golden.py [case ...]      (build container only; ~1 h for all)

Stored per case (tests/golden/fullsize/<case>.pt):
  forward   loss, acc, the arg-max label map and the top-1 / top-2 margin of EVERY pixel, the 150 log-probabilities of a seeded
            pixel sample (all pixels when there are <= 8192);
  backward  `anchor_grads_enc / _dec`: make_golden.anchor() of EVERY parameter gradient (float64 value in full for tensors of
            <= 4096 elements, else a seeded 1024-element sample; the reference's fp32 reproducibility band around it);
  state     `anchor_after_enc / _dec`: the same for every BN running statistic after the step (a parameter after the step is
            w - lr (g + wd w): it inherits its gradient's record, tests/util.post_step_bands), `loss64`."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                      # noqa: E402  (puts /root/reference first on sys.path, asserts the import is the reference's)

CASES = {
    'cfg1_r50d_ppmds_512': ('resnet50dilated', 'ppm_deepsup', 2048, 0.4, 8, 512, 512),
    'cfg2_r50_upernet_512': ('resnet50', 'upernet', 2048, None, 4, 512, 512),
    'cfg3_r101d_ppmds_376x504': ('resnet101dilated', 'ppm_deepsup', 2048, 0.4, 8, 376, 504),
    'cfg3_r101d_ppmds_456x680': ('resnet101dilated', 'ppm_deepsup', 2048, 0.4, 8, 456, 680),
    'cfg4_hrnetv2_c1_512': ('hrnetv2', 'c1', 720, None, 4, 512, 512),
}
SAMPLE = 8192


def pixel_sample(npix, k=SAMPLE):
    """seeded pixel sample shared with the test (tests/test_gpu_models.py rebuilds it): all pixels when there are <= k"""
    if npix <= k:
        return torch.arange(npix)
    return torch.randperm(npix, generator=torch.Generator().manual_seed(npix))[:k].sort()[0]


def run(case):
    arch_enc, arch_dec, fc_dim, dss, rate, H, W = CASES[case]
    r = MG.run_case(case, arch_enc, arch_dec, fc_dim, 2, H, W, rate, True, dss, step=True, seed=3)
    pred = r['pred']
    n, c, h, w = pred.shape
    rows = pred.permute(0, 2, 3, 1).reshape(n * h * w, c)
    top2 = rows.topk(2, dim=1)[0]
    idx = pixel_sample(n * h * w)
    stat = lambda recs: {k: v for k, v in recs.items() if k.rsplit('.', 1)[-1] in ('running_mean', 'running_var')}   # noqa: E731
    return {'meta': dict(case=case, arch_encoder=arch_enc, arch_decoder=arch_dec, fc_dim=fc_dim, deep_sup_scale=dss, seg_rate=rate,
                         n=2, h=H, w=W, seed=3, seed_weights=(3, 4), seed_dropout=(3, 4), seed_batch=307, lr=0.02,
                         torch=torch.__version__, pred_shape=list(pred.shape), variants=('default',) + tuple(MG.VARIANTS)),
            'loss': r['loss'], 'acc': r['acc'], 'loss64': r['loss64'],
            'argmax': rows.argmax(1).to(torch.int16).reshape(n, h, w).clone(),
            'margin': (top2[:, 0] - top2[:, 1]).reshape(n, h, w).clone(),
            'pixels': idx.clone(), 'logp': rows[idx].clone(),
            'anchor_grads_enc': r['anchor_grads_enc'], 'anchor_grads_dec': r['anchor_grads_dec'],
            'anchor_after_enc': stat(r['anchor_after_enc']), 'anchor_after_dec': stat(r['anchor_after_dec'])}


def main():
    torch.set_num_threads(os.cpu_count())
    out_dir = os.path.join(HERE, 'fullsize')
    os.makedirs(out_dir, exist_ok=True)
    for case in (sys.argv[1:] or sorted(CASES)):
        t0 = time.time()
        r = run(case)
        path = os.path.join(out_dir, case + '.pt')
        torch.save(r, path)
        ng = len(r['anchor_grads_enc']) + len(r['anchor_grads_dec'])
        print('%-26s %7.1f KB loss=%.6f (float64 %.6f) acc=%.6f min margin %.2e, %d gradient tensors, %.0f s'
              % (case, os.path.getsize(path) / 1024, r['loss'].item(), r['loss64'], r['acc'].item(), r['margin'].min().item(), ng,
                 time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
