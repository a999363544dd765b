"""Generate tests/golden/metrics_golden.npz with the UNMODIFIED reference metric functions
(/root/reference/mit_semseg/utils.py:128-156).  Build container only:

    python tests/golden/make_metrics_golden.py
"""
import importlib.util
import os

import numpy as np

REF = os.environ.get('SEMSEG_REFERENCE', '/root/reference')


def _load_reference():
    spec = importlib.util.spec_from_file_location('ref_utils', os.path.join(REF, 'mit_semseg', 'utils.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def synth_case(name, C, H, W, seed):
    """seeded inputs (numpy RandomState streams are stable across numpy versions); shared with the tests"""
    rng = np.random.RandomState(seed)
    scores = rng.rand(C, H, W).astype(np.float32)
    scores[:, ::5, ::3] = scores[:1, ::5, ::3]                 # exact ties: the first maximum must win
    label = rng.randint(-1, C, size=(H, W)).astype(np.int64)
    if name == 'allignored':
        label[:] = -1
    return scores, label


def main():
    ref = _load_reference()
    out = {}
    cases = [('a', 150, 37, 53, 0), ('b', 150, 64, 64, 1), ('c', 21, 9, 7, 2), ('d', 150, 128, 96, 3), ('allignored', 150, 8, 8, 4),
             ('perfect', 150, 16, 16, 5)]
    for name, C, H, W, seed in cases:
        scores, label = synth_case(name, C, H, W, seed)
        pred = np.argmax(scores, axis=0).astype(np.int64)          # == torch.max(dim) first-max rule (checked in the test)
        if name == 'perfect':
            label = pred.copy()
        acc, pix = ref.accuracy(pred, label)
        inter, union = ref.intersectionAndUnion(pred, label, C)
        # inputs are regenerated from the seed by the tests (synth_case); only the reference's OUTPUTS are stored
        out.update({name + '_cfg': np.array([C, H, W, seed]), name + '_pred': pred.astype(np.int16), name + '_acc': np.float64(acc),
                    name + '_pix': np.int64(pix), name + '_inter': inter.astype(np.int64), name + '_union': union.astype(np.int64)})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'metrics_golden.npz'), **out)
    print('wrote metrics_golden.npz', sorted(k for k in out if k.endswith('_acc')))


if __name__ == '__main__':
    main()
