"""Per-kernel difference of two rocprofv3 kernel-stats tables of the SAME tree (tools/rocprof_summary.py output): which kernels carry
the step-time difference between two boxes (round-3 review: the same tree ran 8 % apart on a "slow" and a "fast" box with the
dominant GEMM taking the same time on both).

    python tools/kernel_stats_diff.py slow.csv fast.csv [steps_A [steps_B]]     # ms per step per kernel family, sorted by difference
steps default: calls of nll_acc_finish_kernel / 2 (two loss heads per step of the PPM_deepsup configurations)"""
import csv
import re
import sys


def family(name):
    name = re.sub(r'\(.*', '', name).replace('void ', '').strip()
    for key, fam in (('igemm_dma', 'GEMM fwd/dgrad (LDS-DMA)'), ('igemm_rs', 'GEMM fwd/dgrad (register staged)'),
                     ('wino_fused', 'Winograd fused dgrad'), ('wgrad_dma', 'weight gradient (LDS-DMA)'), ('wgrad_taps', 'weight gradient (all taps)'),
                     ('wgrad_kernel', 'weight gradient (register staged)'), ('wgrad_multi', 'weight gradient (register staged)'), ('split_wgrad_reduce', 'split reduces'),
                     ('split_gemm_reduce', 'split reduces'), ('bn_apply', 'BN apply fwd'), ('bn_bwd_apply', 'BN apply bwd'),
                     ('bn_bwd_mm_partial', 'BN partial sums bwd'), ('bn_stats_mm_partial', 'BN statistics sweep'),
                     ('finish_fused', 'BN finish kernels'), ('wino_', 'Winograd transforms'), ('wprep', 'weight preparation'),
                     ('sgd', 'SGD'), ('add_act', 'gradient sums at forks'), ('copy', 'copies'), ('pool', 'pooling'),
                     ('bilinear', 'bilinear'), ('softmax', 'head'), ('nll', 'head')):
        if key in name:
            return fam
    return 'other'


def load(path, steps):
    fams, names = {}, {}
    rows = list(csv.DictReader(open(path)))
    if not steps:
        steps = sum(int(r['Calls']) for r in rows if 'nll_acc_finish_kernel' in r['Name']) / 2.0 or 16.0
    for r in rows:
        ms = int(r['TotalDurationNs']) / 1e6 / steps
        fams[family(r['Name'])] = fams.get(family(r['Name']), 0.0) + ms
        names[r['Name'][:70]] = ms
    return fams, names, steps


def main():
    a, b = sys.argv[1], sys.argv[2]
    sa = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    sb = float(sys.argv[4]) if len(sys.argv) > 4 else sa
    fa, na, sa = load(a, sa)
    fb, nb, sb = load(b, sb)
    print('ms per step by kernel family: A = %s (%g steps), B = %s (%g steps)' % (a, sa, b, sb))
    print('%-36s %8s %8s %8s %7s' % ('family', 'A', 'B', 'A - B', 'A / B'))
    rows = sorted(set(fa) | set(fb), key=lambda k: -(fa.get(k, 0) - fb.get(k, 0)))
    for k in rows:
        x, y = fa.get(k, 0.0), fb.get(k, 0.0)
        print('%-36s %8.3f %8.3f %+8.3f %7.3f' % (k, x, y, x - y, x / y if y else float('nan')))
    print('%-36s %8.3f %8.3f %+8.3f %7.3f' % ('sum', sum(fa.values()), sum(fb.values()), sum(fa.values()) - sum(fb.values()),
                                                sum(fa.values()) / sum(fb.values())))
    print('\nlargest single kernels by difference:')
    for k in sorted(set(na) & set(nb), key=lambda k: -(na[k] - nb[k]))[:12]:
        print('  %-70s %7.3f %7.3f %+7.3f  x%.3f' % (k, na[k], nb[k], na[k] - nb[k], na[k] / nb[k] if nb[k] else float('nan')))


if __name__ == '__main__':
    main()
