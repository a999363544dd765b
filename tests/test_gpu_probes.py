"""Box probes (csrc/probe.hip) and the timeline markers of a captured training step (mit_semseg/scaling_model.TimelineProbe):
what bench.py's `box` and `scaling_model` blocks are built from."""
import ctypes
import json
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_probe_kernels_run_and_report_sane_rates():
    from mit_semseg import _native
    L = _native.lib()
    vp = ctypes.c_void_p
    dev = torch.device('cuda:0')
    st = vp(torch.cuda.current_stream().cuda_stream)
    blocks, iters = 512, 2000
    sink = torch.zeros(4, device=dev)
    cyc = torch.zeros(blocks, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _native.check(L.semseg_probe_mfma_f16(vp(sink.data_ptr()), blocks, iters, vp(cyc.data_ptr()), st), 'mfma')
    torch.cuda.synchronize()
    e0.record()
    _native.check(L.semseg_probe_mfma_f16(vp(sink.data_ptr()), blocks, iters, vp(cyc.data_ptr()), st), 'mfma')
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    tflops = blocks * 4.0 * iters * 8 * 2 * 32 * 32 * 16 / t * 1e-12
    ghz = float(cyc.double().mean().item()) / t * 1e-9
    print('mfma probe: %.0f TFLOP/s, %.2f GHz' % (tflops, ghz))
    assert 300 < tflops < 2600 and 0.8 < ghz < 2.6 and int(cyc.min().item()) > 0
    a = torch.randn(1 << 22, device=dev)
    b = torch.zeros_like(a)
    _native.check(L.semseg_probe_copy(vp(a.data_ptr()), vp(b.data_ptr()), a.numel() * 4, st), 'copy')
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert L.semseg_probe_copy(vp(a.data_ptr()), vp(b.data_ptr()), 24, st) != 0          # not a multiple of 16 bytes
    slots = torch.zeros(2, dtype=torch.int64, device=dev)
    _native.check(L.semseg_probe_timestamp(vp(slots.data_ptr()), st), 'ts')
    _native.check(L.semseg_probe_empty(st), 'empty')
    _native.check(L.semseg_probe_timestamp(vp(slots.data_ptr() + 8), st), 'ts')
    torch.cuda.synchronize()
    t0, t1 = slots.tolist()
    assert 0 < t0 < t1


def test_timeline_markers_inside_a_replayed_step_are_ordered():
    from mit_semseg.models import ModelBuilder, SegmentationModule, resnet
    from mit_semseg.models.models import ResnetDilated
    from mit_semseg.engine import TrainStep
    from mit_semseg import scaling_model as smod
    from oracle import semseg_oracle as O
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    enc = ResnetDilated(resnet.resnet18(pretrained=False), dilate_scale=8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=512, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), 0.4).to(dev).train()
    img, lab = O.synth_batch(2, 64, 64, 8, seed=5)
    feed = {'img_data': img.to(dev), 'seg_label': lab.to(dev)}
    step = TrainStep(sm, max_iters=1000, graph=True)
    for _ in range(3):
        step.step(feed)
    probe = smod.TimelineProbe(list(enc.parameters()) + list(dec.parameters()), bucket_bytes=8 << 20)
    assert len(probe.groups) >= 3 and sum(probe.bucket_bytes) == 4 * sum(p.numel() for p in sm.parameters())
    step.timeline = probe
    step._graphs.clear()
    for _ in range(3):
        loss, _ = step.step(feed)
    torch.cuda.synchronize()
    assert step.stats['captured'] == 2 and torch.isfinite(loss)
    first = probe.read()
    step.step(feed)
    torch.cuda.synchronize()
    ticks = probe.read()
    step.timeline = None
    probe.detach()
    # autograd does not finish the buckets strictly in bucket order (it walks the graph, not the parameter list): every bucket
    # lies inside backward, the phases are ordered
    buckets = [ticks['bucket%d' % i] for i in range(len(probe.groups))]
    assert ticks['step_begin'] < ticks['fwd_end'] <= min(buckets) and max(buckets) <= ticks['bwd_end'] < ticks['step_end'], ticks
    assert buckets[0] < buckets[-1]                           # the classifier's bucket long before the stem's
    assert ticks['step_begin'] > first['step_end']            # a replay re-stamps every marker
    line = smod.model_line(10.0, ticks, probe.bucket_bytes, smod.syncbn_payloads(sm))
    json.dumps(line)
    assert line['predicted']['8']['peer_exchange']['ms_per_step'] >= 10.0
