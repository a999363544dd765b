#!/usr/bin/env python3
"""ISA lint of the gfx950 kernels: full-drain waits (`s_waitcnt vmcnt(0)`) the compiler put in front of stores / inside loops.

Why: on gfx9 the vector-memory counter counts loads AND stores, so a `vmcnt(0)` in front of a store also waits for the PREVIOUS
store's acknowledgement.  The compiler (SIInsertWaitcnts) emits one whenever a value that MAY still be in flight is used in a
block it cannot prove was preceded by a wait on every path -- e.g. a conditionally loaded value (a bias that is only read when
the pointer is non-null) used inside per-row conditional blocks (row < M): every row's store then leaves one memory round trip
after the previous one.  That is what the GEMM epilogue of csrc/conv_split.hip did until round 6 (69 full drains for 73 stores in
the 256 x 256 tile; one unconditional use of the loaded value in front of the rows -- `asm volatile("" : "+v"(x))` -- removes
them: 5 drains for 73 stores, configs[1] -1.2 %, configs[4] -2.6 % in-box).

Usage:  python tools/isa_wait_lint.py [csrc/file.hip ...]      (default: every csrc/*.hip; ~1 min for conv_split.hip)
Prints, per kernel with >= --min full drains: drains, stores, loads.  Exit code 0 always (a report, not a gate)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def asm_of(src, out):
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
           '-Wno-unused-result', '--cuda-device-only', '-S', src, '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def scan(path):
    rows, name = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\S+):\s+; @', line)
        if m:
            name = m.group(1)
            rows[name] = [0, 0, 0]
            continue
        if name is None:
            continue
        if 's_waitcnt vmcnt(0)' in line:
            rows[name][0] += 1
        elif re.search(r'\b(global|buffer|flat)_store', line):
            rows[name][1] += 1
        elif re.search(r'\b(global|buffer|flat)_load', line):
            rows[name][2] += 1
    return rows


def short(name):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = name
    out = re.sub(r'\(.*', '', out)
    return out[:150]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='*')
    ap.add_argument('--min', type=int, default=8, help='report kernels with at least this many full drains')
    a = ap.parse_args()
    files = a.files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    with tempfile.TemporaryDirectory() as d:
        with ThreadPoolExecutor(4) as ex:
            outs = list(ex.map(lambda f: asm_of(f, os.path.join(d, os.path.basename(f) + '.s')), files))
        for f, o in zip(files, outs):
            rows = scan(o)
            bad = sorted(((v, k) for k, v in rows.items() if v[0] >= a.min), reverse=True)
            print('== %s: %d kernels, %d with >= %d full drains' % (os.path.basename(f), len(rows), len(bad), a.min))
            for (w, s, l), k in bad:
                print('  drains %3d  stores %3d  loads %3d  %s' % (w, s, l, short(k)))
    return 0


if __name__ == '__main__':
    sys.exit(main())
