"""ade20k_train_sizes.npz: the (width, height) histogram of the reference's data/training.odgt (20 210 records, 3 972 distinct
sizes).  bench.py / the tests draw the per-GPU batch shapes of BASELINE configs[3] (multi-scale variable-size batches,
dataset.py:121-142) from it -- synthetic pixels, real shape statistics.  Build container only:

    python tests/golden/make_sizes_golden.py
"""
import collections
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('SEMSEG_REFERENCE', '/root/reference')


def main():
    c = collections.Counter()
    for line in open(os.path.join(REF, 'data', 'training.odgt')):
        r = json.loads(line)
        c[(int(r['width']), int(r['height']))] += 1
    items = sorted(c.items())
    np.savez_compressed(os.path.join(HERE, 'ade20k_train_sizes.npz'),
                        width=np.array([k[0] for k, _ in items], np.int16), height=np.array([k[1] for k, _ in items], np.int16),
                        count=np.array([v for _, v in items], np.int32))
    print(len(items), 'distinct sizes,', sum(c.values()), 'records')


if __name__ == '__main__':
    main()
