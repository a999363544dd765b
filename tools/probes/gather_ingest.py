"""Operand ingest of a CU: GB/s per CU at which 4-wave blocks gather 1 KiB pieces of (1024 / SEG) rows x SEG bytes from a pitched
matrix -- the shape of every operand fetch of the GEMM kernels (SEG = 64: the 32-deep k-tiles, 128: the 64-deep ones) -- by LDS-DMA and by plain
loads into registers, for one and two blocks per CU and for a matrix that fits the L2s / the Infinity Cache / neither.
    python tools/probes/gather_ingest.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
from mit_semseg import _native  # noqa: E402


def main():
    L = _native.lib()
    dev = torch.device('cuda:0')
    vp = ctypes.c_void_p
    sink = torch.zeros(4, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    print('GB/s per CU (and TB/s of the chip) of 1 KiB operand pieces = (1024 / SEG) rows x SEG bytes, 4 steps of 32 KiB in flight per block')
    for mb, what in ((16, 'fits the L2s'), (128, 'fits the Infinity Cache'), (1024, 'HBM')):
        pitch = 4096 + 256
        rows = (mb << 20) // pitch // 2048 * 2048
        src = torch.empty(rows * pitch, dtype=torch.uint8, device=dev)
        src.zero_()
        for per_cu in (1, 2):
            blocks = cus * per_cu
            steps = 4096 // per_cu
            row = []
            for dma in (1, 0):
                for seg in (32, 64, 128, 256, 1024):
                    if pitch % seg:
                        row.append('   n/a')
                        continue
                    run = lambda: _native.check(L.semseg_probe_gather(vp(src.data_ptr()), rows, pitch, seg, dma, blocks, steps,   # noqa: E731
                                                                       vp(sink.data_ptr()), st), 'probe_gather')
                    run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 3
                    gbs = blocks * steps * 32768 / ms * 1e-6
                    row.append('%s%d: %5.1f (%4.1f)' % ('dma' if dma else 'reg', seg, gbs / cus, gbs * 1e-3))
            print('%5d MB (%s), %d block(s) per CU:  %s' % (mb, what, per_cu, '  '.join(row)))
        # how many steps (32 KiB each) a block keeps in flight: throughput = bytes in flight / latency until the path saturates
        for seg in (64, 128):
            row = []
            for d in (1, 2, 4, 7):
                run = lambda: _native.check(L.semseg_probe_gather(vp(src.data_ptr()), rows, pitch, seg, 10 + d, cus, 4096,   # noqa: E731
                                                                   vp(sink.data_ptr()), st), 'probe_gather')
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                row.append('%d x 32 KiB in flight: %5.1f' % (d, cus * 4096 * 32768 / ms * 1e-6 / cus))
            print('%5d MB, LDS-DMA SEG %d, one block per CU:  %s' % (mb, seg, '  '.join(row)))
        del src


if __name__ == '__main__':
    main()
