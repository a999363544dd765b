"""Build libsemseg_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a
plain C ABI (include/semseg_hip.h).  Usage: python build_native.py [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, 'mit_semseg', '_native')
LIB = os.path.join(OUT_DIR, 'libsemseg_hip.so')
SOURCES = ['conv_igemm.hip', 'conv_wgrad.hip', 'conv_split.hip', 'weights_prep.hip', 'winograd.hip', 'bn.hip', 'pool_resize.hip', 'head.hip', 'input_pipeline.hip', 'depthwise.hip', 'grouped.hip', 'comm.hip', 'peer.hip', 'probe.hip', 'batch.hip', 'api.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
         '-Wno-unused-result']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _depfile_deps(depfile):
    """prerequisites of a compiler-written depfile (`hipcc -MMD -MF`: Makefile syntax, one rule per offload pass); None when
    the file is missing or unreadable -- the object is then rebuilt"""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    deps = set()
    for rule in text.replace('\\\n', ' ').splitlines():
        if ':' not in rule:
            continue
        for tok in rule.split(':', 1)[1].split():
            deps.add(tok)
    return sorted(deps)


def _stale(src, obj):
    """an object is rebuilt when its source, any user header the compiler saw last time (the -MMD depfile next to it) or
    this build script itself (its flags live here) is newer than the object"""
    if _newer(src, obj) or _newer(os.path.abspath(__file__), obj):
        return True
    deps = _depfile_deps(obj + '.d')
    if deps is None:
        return True
    return any((not os.path.exists(d)) or _newer(d, obj) for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force=False, verbose=True):
    """Compile what is out of date and link.  Safe under concurrent callers (one process per GPU all call
    __graft_entry__.build()): an exclusive file lock serialises them, the freshness check runs under the lock (later callers
    find everything built), and every output is written to a temporary name and renamed into place, so a process that has
    already mapped the library never sees it truncated."""
    import fcntl
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    objs, jobs = [], []
    tag = '.tmp%d' % os.getpid()
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(src, obj):
            # -MMD: the compiler lists every user header it read (csrc/*.h, include/semseg_hip.h) in obj.d; no hand-kept list
            jobs.append((obj, [HIPCC] + FLAGS + ['-MMD', '-MF', obj + '.d' + tag, '-c', src, '-o', obj + tag]))
    if jobs:
        def run(job):
            obj, cmd = job
            _run(cmd, verbose)
            os.replace(obj + '.d' + tag, obj + '.d')
            os.replace(obj + tag, obj)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        _run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + tag] + objs + ['-ldl'], verbose)
        os.replace(LIB + tag, LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
