"""Floor of ONE training step in its CURRENT decomposition into launches, from the PMC passes of tools/gpu_pmc_step.sh and the call
trace of tools/abi_call_trace.py:  per launch  max(MFMA time of the executed products at the clock the dense kernels hold,
algorithmic operand bytes at the achievable HBM rate, a launch floor)  next to the measured duration.  It answers "which launches are
far from what the hardware allows" without another GPU run.
    python tools/abi_call_trace.py > /tmp/abi_calls.txt;  python tools/step_floor_model.py gpurun_out/<tag> /tmp/abi_calls.txt
Constants: 2.5 PFLOP/s dense 16-bit MFMA at 2.4 GHz scaled to the 1.7 GHz the MFMA-dense kernels run at under load,
5.5 TB/s for streaming kernels (what add / BN-apply / SGD reach here), 2.5 us per launch inside a hipGraph.
The batched small weight gradients (wgrad_multi_kernel, one launch per 24 layers) carry no per-layer geometry in the call trace: they are
priced by the bytes they moved only (their MFMA time is a few per cent of the launch).
Round 6: a side-by-side launch (csrc/batch.h many_kernel) carries several convolutions, so the one-to-one match of launches and
C-ABI calls this model needs only holds with SEMSEG_BATCH_BRANCHES=0 (tools/gpu_pmc_step.sh sets nothing: run it that way for the floor)."""
import csv
import glob
import os
import re
import sys

MFMA = 2.5e15 * 1.7 / 2.4
HBM = 5.5e12
LAUNCH = 2.5e-6
CONV_CALLS = ('conv2d_dgrad_h2', 'conv2d_fwd_h2', 'conv2d_fwd_stats_h2', 'conv2d_wgrad_h2', 'conv2d_wgrad_slabs_h2', 'winograd_gemm_h2', 'winograd_wgrad_gemm_h2',
              'winograd_gemm_output_h2')


def short(n):
    n = re.sub(r'^void ', '', n)
    m = re.match(r'semseg_batch::(one|many)_kernel<(\w+?)_body\b(.*)', n)      # csrc/batch.h: name the body the generic kernel runs
    if m:
        n = m.group(2) + m.group(3)
    return re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', n)[:56]


def counters(root, sub, counter):
    d = {}
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                i = int(r['Dispatch_Id'])
                d[i] = d.get(i, 0.0) + float(r['Counter_Value'])
    return d


def main():
    root, calls_path = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_step_summary import last_period
    F, W = counters(root, 'fetch', 'FETCH_SIZE'), counters(root, 'write', 'WRITE_SIZE')
    tr = {}
    for f in glob.glob(os.path.join(root, 'fetch', '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            tr[int(r['Dispatch_Id'])] = r
    ids = sorted(F)
    p, tail = last_period([short(tr[i]['Kernel_Name']) for i in ids])
    step = ids[len(ids) - tail - p:len(ids) - tail]
    calls = [l.split() for l in open(calls_path) if l.split() and l.split()[0] in CONV_CALLS]
    convs = [i for i in step if short(tr[i]['Kernel_Name']).startswith(('igemm', 'wgrad_kernel', 'wgrad_dma', 'wgrad_taps', 'wino_fused'))]
    assert len(convs) == len(calls), (len(convs), len(calls))
    geom = dict(zip(convs, calls))
    rows = []
    for i in step:
        r = tr[i]
        name = short(r['Kernel_Name'])
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
        moved = (2 * F[i] + W.get(i, 0.0)) * 1024
        flops, alg, label = 0.0, moved, ''
        if i in geom:
            c = geom[i]
            a = [int(v) for v in c[1:]]
            if c[0].startswith('conv2d'):
                if c[0] not in ('conv2d_wgrad_h2', 'conv2d_wgrad_slabs_h2'):
                    a = a[1:]                       # leading dimension of the output
                n, h, w, ci, k, rr, ss, st, pad, dil = a[:10]
                oh, ow = (h + 2 * pad - dil * (rr - 1) - 1) // st + 1, (w + 2 * pad - dil * (ss - 1) - 1) // st + 1
                flops = 3 * 2.0 * n * oh * ow * k * ci * rr * ss
                x, y, wt = n * h * w * ci * 4, n * oh * ow * k * 4, ci * k * rr * ss * 4
                alg = {'conv2d_fwd_stats_h2': x + wt + y, 'conv2d_fwd_h2': x + wt + y, 'conv2d_dgrad_h2': y + wt + x,
                       'conv2d_wgrad_h2': x + y + wt, 'conv2d_wgrad_slabs_h2': x + y + wt}[c[0]]
                label = '%s %dx%dx%d %d->%d %dx%d s%d d%d' % (c[0][7:-3].replace('_slabs', ''), n, h, w, ci, k, rr, ss, st, dil)
            elif c[0] == 'winograd_gemm_output_h2':  # fused GEMM + output transform: z_ld, N, H, W, C (reduction), K, dil, form
                zld, n, h, w, ci, k, dil = a[:7]
                tiles = n * dil * dil * ((-(-h // dil) + 1) // 2) * ((-(-w // dil) + 1) // 2)
                flops = 3 * 2.0 * 16 * tiles * ci * k
                alg = 16 * 4 * (tiles * ci + ci * k) + 4 * n * h * w * k
                label = 'winograd_gemm_output tiles %d %d->%d' % (tiles, ci, k)
            else:                                   # batched Winograd GEMMs: tiles, C, K (16 positions)
                tiles, ci, k = a[:3]
                flops = 3 * 2.0 * 16 * tiles * ci * k
                alg = 16 * 4 * (tiles * ci + ci * k + tiles * k)
                label = '%s tiles %d %d->%d' % (c[0][:-3], tiles, ci, k)
        floor = max(flops / MFMA, alg / HBM, LAUNCH)
        bound = 'mfma' if flops / MFMA >= max(alg / HBM, LAUNCH) else ('hbm' if alg / HBM >= LAUNCH else 'launch')
        rows.append((name, label, dur, floor, bound, moved, alg))
    tot_d, tot_f = sum(r[2] for r in rows), sum(r[3] for r in rows)
    print('launches %d   measured (under PMC, serialised) %.2f ms   floor of this decomposition %.2f ms' % (len(rows), tot_d * 1e3, tot_f * 1e3))
    for b in ('mfma', 'hbm', 'launch'):
        sel = [r for r in rows if r[4] == b]
        print('  bound by %-6s: %4d launches, measured %6.2f ms, floor %6.2f ms' % (b, len(sel), sum(r[2] for r in sel) * 1e3, sum(r[3] for r in sel) * 1e3))
    agg = {}
    for name, label, dur, floor, bound, moved, alg in rows:
        key = (name, label)
        e = agg.setdefault(key, [0, 0.0, 0.0, bound, 0.0, 0.0])
        e[0] += 1; e[1] += dur; e[2] += floor; e[4] += moved; e[5] += alg
    print('\n%-56s %-44s %4s %9s %9s %8s %6s %9s %9s' % ('kernel', 'geometry', 'n', 'meas us', 'floor us', 'gap us', 'bound', 'moved MB', 'alg MB'))
    for (name, label), (n, dur, floor, bound, moved, alg) in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:70]:
        print('%-56s %-44s %4d %9.1f %9.1f %8.1f %6s %9.1f %9.1f' % (name, label, n, dur * 1e6, floor * 1e6, (dur - floor) * 1e6, bound, moved / 1e6, alg / 1e6))


if __name__ == '__main__':
    main()
