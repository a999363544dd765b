"""Where a block of igemm_dma_kernel spends its time: entry -> first k-tile landed -> k loop done -> epilogue done, from wall-clock stamps
(100 MHz) that a -DSEMSEG_STAMPS build of csrc/conv_split.hip leaves per block.  The stamped library is built BESIDE the product one:

    python tools/probes/gemm_phase_stamps.py --build            # here (no GPU): build_ab/stamps/libsemseg_hip.so
    gpurun -- 'python tools/probes/gemm_phase_stamps.py'        # on the box: layer4's dilated 3x3 forward (Winograd GEMM, tile 12)
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')
OUT = os.path.join(ROOT, 'build_ab', 'stamps')
LIB = os.path.join(OUT, 'libsemseg_hip.so')


VARIANTS = {'': [], '_nomma': ['-DSEMSEG_PROBE_NOMMA'], '_nodma': ['-DSEMSEG_PROBE_NODMA'],
            '_dmaonly': ['-DSEMSEG_PROBE_NOMMA', '-DSEMSEG_PROBE_NOREAD'], '_mmaonly': ['-DSEMSEG_PROBE_NODMA', '-DSEMSEG_PROBE_NOREAD']}


def build():
    """the stamped library and its four timing variants (wrong results: the loop without its MFMAs / without its DMA / DMA and barriers
    only / MFMAs and barriers only)"""
    sys.path.insert(0, PKG)
    import build_native as B
    from concurrent.futures import ThreadPoolExecutor
    B.build()
    os.makedirs(OUT, exist_ok=True)

    def one(item):
        name, flags = item
        obj = os.path.join(OUT, 'conv_split%s.o' % name)
        subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DSEMSEG_STAMPS'] + flags + ['-c', os.path.join(B.CSRC, 'conv_split.hip'), '-o', obj])
        objs = [obj if s == 'conv_split.hip' else os.path.join(B.OUT_DIR, s.replace('.hip', '.o')) for s in B.SOURCES]
        lib = LIB.replace('.so', '%s.so' % name)
        subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl'])
        return lib
    with ThreadPoolExecutor(5) as ex:
        for lib in ex.map(one, VARIANTS.items()):
            print(lib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--layer', default='l4_conv2_d4')
    ap.add_argument('--variant', default='', choices=sorted(VARIANTS))
    ap.add_argument('--wino-tile', type=int, default=-1, help='pin this tile for the batched Winograd GEMM (plan pass 3) instead of the tuned one')
    a = ap.parse_args()
    if a.build:
        return build()
    global LIB
    LIB = LIB.replace('.so', '%s.so' % a.variant)
    os.environ['SEMSEG_NATIVE_LIB'] = LIB
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import numpy as np
    import torch
    import torch.nn.functional as F
    from mit_semseg import ops, _native
    import conv_bench
    L = _native.lib()
    raw = ctypes.CDLL(LIB)
    raw.semseg_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    layer = [l for l in conv_bench.LAYERS if l[0] == a.layer][0]
    _, n, c, h, w, k, ks, st, pad, dil, _ = layer
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(k, c, ks, ks, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(k).to(dev).train()
    ops.CONV_MODE = 'h2'

    # the product's input of this layer is the output of a fused conv -> BN -> ReLU node (it carries its h2 planes and bounds)
    w0 = torch.nn.Parameter((torch.randn(c, c, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
    bn0 = torch.nn.BatchNorm2d(c).to(dev).train()
    ops.prepare_conv_weights([w0])
    x = ops.conv_bn_act(x, w0, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, bn0.num_batches_tracked, None, 1, 0, 1,
                        training=True, relu=True)

    if a.wino_tile >= 0:
        from mit_semseg import tuner
        tiles = n * ((h + 1) // 2) * ((w + 1) // 2)
        geom = (tiles, 1, 1, c, k, 3, 3, 1, 1, 1)
        tuner._done[('h2', 3) + geom] = (a.wino_tile, 1, None)
        print('pinned wino GEMM tile %d: rc %d' % (a.wino_tile, L.semseg_conv2d_h2_set_plan(3, *geom, a.wino_tile, 1)))

    def run():
        # the fused conv -> BN -> ReLU node is the caller of the Winograd forward in the product
        return ops.conv_bn_act(x, wt, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, None, st, pad, dil,
                               training=True, relu=True)

    wt = torch.nn.Parameter(wt)          # the training path of the fused node, on weight planes prepared as the engine prepares them
    ops.prepare_conv_weights([wt])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if 'igemm' in e.key or 'wino' in e.key:
            print('   kernel %-70s x%d  %.1f us' % (e.key[:70], e.count, e.device_time_total / max(1, e.count)))
    nblk = 16384
    buf = (ctypes.c_ulonglong * (4 * nblk))()
    rc = raw.semseg_debug_stamps(buf, 4 * nblk)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 4).astype(np.int64)
    print('rc %d; blocks with stamp i set: %s; lib %s' % (rc, [(s[:, i] > 0).sum() for i in range(4)], _native.lib()._name))
    live = (s[:, 3] > 0) & (s[:, 1] > 0)
    s = s[live]
    if not len(s):
        return
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    print('rc %d, %d blocks stamped (the LAST igemm_dma_kernel launch of the pass that ran through the final epilogue path)' % (rc, len(s)))
    print('entry              : min %.1f  median %.1f  max %.1f us after the first block entered' % (us[:, 0].min(), np.median(us[:, 0]), us[:, 0].max()))
    for i, name in ((1, 'entry -> first tile landed'), (2, 'k loop'), (3, 'epilogue')):
        d = us[:, i] - us[:, i - 1]
        print('%-26s: min %.1f  median %.1f  max %.1f us' % (name, d.min(), np.median(d), d.max()))
    print('block lifetime            : median %.1f us; launch span (first entry -> last exit) %.1f us' % (np.median(us[:, 3] - us[:, 0]), us[:, 3].max()))


if __name__ == '__main__':
    main()
