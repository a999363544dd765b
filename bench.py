"""Benchmark of the MI355X hot path on BASELINE.json's metric: training images/sec at 512x512, bs 2 per GPU,
ade20k-resnet50dilated-ppm_deepsup (configs[1]), synthetic ADE20K-shaped batches, seeded random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --config 2|3|4      # the other BASELINE.json configs, same JSON line (default 1 = the metric's config):
                                        # 2 R50+UPerNet, 3 R101dilated+PPM_deepsup on variable-size batches, 4 HRNetV2+C1

One "step" = train.py:34-48: zero_grad, SegmentationModule.forward (loss+acc), backward, 2x SGD (poly LR), fp32.
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch            # noqa: E402
import torch.nn as nn   # noqa: E402

# SURVEY 8d (hooks on the reference modules): 2*MACs of every Conv2d; train = 3*fwd - fwd(first conv)
# BASELINE.json configs[i] -> workload.  train GFLOP/img at 512x512 (variable-size batches scale it by H*W / 512^2);
# `dominant`: the largest single conv launch of the step -- (pass 0 fwd / 1 dgrad, (N,H,W,C,K,R,S,stride,pad,dil), layer) --
# timed live for the roofline entry.  conv_last of the PPM / UPerNet heads runs its FORWARD through the Winograd path, so the
# largest single launch of those steps is its data gradient.
CONFIGS = {
    1: dict(name='resnet50dilated+ppm_deepsup', yaml='ade20k-resnet50dilated-ppm_deepsup', enc='resnet50', dilated=True,
            dec='ppm_deepsup', fc_dim=2048, dss=0.4, rate=8, pad=8, gflop=1224.2,
            dominant=(1, (2, 64, 64, 4096, 512, 3, 3, 1, 1, 1), 'decoder.conv_last.0 3x3 4096->512 @64x64 N=2')),
    2: dict(name='resnet50+upernet', yaml='ade20k-resnet50-upernet', enc='resnet50', dilated=False, dec='upernet', fc_dim=2048,
            dss=None, rate=4, pad=32, gflop=1468.0,
            dominant=(1, (2, 128, 128, 2048, 512, 3, 3, 1, 1, 1), 'decoder.conv_last.0.0 3x3 2048->512 @128x128 N=2')),
    3: dict(name='resnet101dilated+ppm_deepsup', yaml='ade20k-resnet101dilated-ppm_deepsup', enc='resnet101', dilated=True,
            dec='ppm_deepsup', fc_dim=2048, dss=0.4, rate=8, pad=8, gflop=1689.6, variable=True,
            dominant=(1, (2, 64, 64, 4096, 512, 3, 3, 1, 1, 1),
                      'decoder.conv_last.0 3x3 4096->512 N=2, timed at the 64x64 map of a 512x512 batch')),
    4: dict(name='hrnetv2+c1', yaml='ade20k-hrnetv2', enc='hrnetv2', dilated=False, dec='c1', fc_dim=720, dss=None, rate=4, pad=32,
            gflop=625.4, dominant=(0, (2, 128, 128, 720, 180, 3, 3, 1, 1, 1), 'decoder.cbr.0 3x3 720->180 @128x128 N=2')),
}
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# HBM-side bytes and the clock under load of the dominant launch come from DATA: profiles/pmc_index.json lists the rocprofv3 PMC
# passes (tools/gpu_pmc.sh; separate runs, FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE) keyed by (mode, pass,
# geometry, tile, split).  The plan the tuner picks in THIS run selects the entry; no entry => traffic null and a warning.
PMC_INDEX = os.path.join(ROOT, 'profiles', 'pmc_index.json')


def pmc_lookup(mode, pass_id, geom, plan, form='direct'):
    try:
        entries = json.load(open(PMC_INDEX))['entries']
    except (OSError, ValueError, KeyError):
        return None
    if plan is None or plan[0] is None:
        return None
    for e in entries:
        if e['mode'] == mode and e['pass'] == pass_id and list(e['geom']) == list(geom) and e.get('form', 'direct') == form and \
                [e['tile'], e['split']] == list(plan[:2]):
            return e
    return None


def algorithmic_bytes(pass_id, geom, mode):
    """bytes one launch has to move at least once: the two operands as split planes (h2: 2 x fp16 = 4 B per element, s3: 6 B,
    f32: 4 B) + the fp32 result"""
    n, h, w, c, k, r, s, stride, pad, dil = geom
    oh, ow = (h + 2 * pad - dil * (r - 1) - 1) // stride + 1, (w + 2 * pad - dil * (s - 1) - 1) // stride + 1
    per = {'h2': 4, 's3': 6, 'f32': 4}[mode]
    act_in, act_out = (n * oh * ow * k, n * h * w * c) if pass_id == 1 else (n * h * w * c, n * oh * ow * k)
    return per * (act_in + c * r * s * k) + 4 * act_out


DTYPE = {'h2': 'f32 (fp32 in/out/accumulate; products on the fp16 MFMA via a scaled 2-way fp16 split, 3 terms, 2^-22 per product)',
         's3': 'f32 (fp32 in/out/accumulate; products on the bf16 MFMA via an exact 3-way bf16 split, 6 terms)', 'f32': 'f32'}
# MFMA products per fp32-accurate MAC block and the issued instruction, per split scheme
SPLIT_TERMS = {'h2': (3, 'v_mfma_f32_32x32x16_f16', '2-way fp16 split, 3 fp16 MFMAs'),
               's3': (6, 'v_mfma_f32_32x32x16_bf16', '3-way bf16 split, 6 bf16 MFMAs')}


def ops_mode():
    from mit_semseg import ops
    return ops.CONV_MODE


def side_by_side_stats():
    """csrc/batch.h: scopes opened, launches recorded and launches the zip issued for them since the process started (the captured
    step is recorded once; replays re-issue what was recorded), direct launches that forced a flush inside a scope"""
    import ctypes
    from mit_semseg import _native, ops
    c = (ctypes.c_longlong * 4)()
    _native.lib().semseg_batch_stats(c)
    return {'enabled': bool(ops.BATCH_BRANCHES), 'scopes': int(c[0]), 'recorded': int(c[1]), 'issued': int(c[2]), 'forced_flushes': int(c[3])}


def launch_plan_provenance():
    """where the (tile, split) launch plans of this run came from: the shipped performance database (plans measured on an MI355X
    by the same tuner, mit_semseg/perfdb), a read-write cache (SEMSEG_TUNE_CACHE), or timed in this process"""
    from mit_semseg import tuner
    return {'from_perfdb': tuner.stats_db['from_perfdb'], 'from_cache': tuner.stats_db['from_cache'],
            'timed_in_this_run': tuner.stats['timed'], 'inherited_by_pixel_bucket': tuner.stats['inherited'],
            'perfdb': os.path.relpath(tuner.PERFDB, ROOT) if (tuner.USE_PERFDB and os.path.exists(tuner.PERFDB)) else None}


def build_model(dev, cfg, seed=304):
    from mit_semseg.models import ModelBuilder, SegmentationModule
    from mit_semseg.models import resnet, hrnet
    from mit_semseg.models.models import ResnetDilated, Resnet
    torch.manual_seed(seed)
    # random init exactly as the reference builds it without a checkpoint (resnet.py:118-124, hrnet.py, models.py:52-61)
    if cfg['enc'] == 'hrnetv2':
        enc = hrnet.hrnetv2(pretrained=False)
    else:
        base = resnet.__dict__[cfg['enc']](pretrained=False)
        enc = ResnetDilated(base, dilate_scale=8) if cfg['dilated'] else Resnet(base)
    dec = ModelBuilder.build_decoder(cfg['dec'], fc_dim=cfg['fc_dim'], num_class=150)
    crit = nn.NLLLoss(ignore_index=-1)
    return SegmentationModule(enc, dec, crit, deep_sup_scale=cfg['dss']).to(dev).train()


def synth_feed(dev, rank, cfg, n=2, h=512, w=512, gen=None):
    g = gen or torch.Generator().manual_seed(304 + rank)
    img = torch.randn(n, 3, h, w, generator=g)
    lab = torch.randint(-1, 150, (n, h // cfg['rate'], w // cfg['rate']), generator=g)
    return {'img_data': img.to(dev), 'seg_label': lab.to(dev)}


def variable_feeds(dev, rank, cfg, count, distinct=0):
    """BASELINE configs[3]: `count` per-GPU batches whose (H, W) follow the multi-scale rule of dataset.py:110-142 (short side
    in {300,...,600}, long side <= 1000, padded to multiples of 8) over the ADE20K size histogram
    (tests/golden/ade20k_train_sizes.npz); one resident synthetic batch per distinct shape.  `distinct` > 0: the stream is
    restricted to its first `distinct` different shapes (in stream order, with the stream's own repetition pattern) -- the
    steady state of a long run, in which the frequent shapes have all been seen."""
    import itertools
    import numpy as np
    from mit_semseg.dataset import batch_shape_stream
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'ade20k_train_sizes.npz'))
    stream = batch_shape_stream(list(zip(d['width'].tolist(), d['height'].tolist())), d['count'], padding_constant=cfg['pad'],
                                seed=304 + rank)
    if distinct:
        keep, shapes = [], []
        for hw in stream:
            if hw not in keep and len(keep) < distinct:
                keep.append(hw)
            if hw in keep:
                shapes.append(hw)
            if len(shapes) == count:
                break
    else:
        shapes = list(itertools.islice(stream, count))
    g = torch.Generator().manual_seed(304 + rank)
    pool = {}
    for hw in shapes:
        if hw not in pool:
            pool[hw] = synth_feed(dev, rank, cfg, h=hw[0], w=hw[1], gen=g)
    return shapes, pool


def dominant_is_winograd(cfg):
    """the dominant launch of the step is the data gradient of the widest 3x3 conv of the head; since round 3 that pass runs in the
    Winograd domain when the layer is eligible (ops.WINOGRAD_DGRAD: 3x3, stride 1, pad == dil, C >= WINOGRAD_MIN_C, K >= 256)"""
    from mit_semseg import ops
    pass_id, geom, _ = cfg['dominant']
    n, h, w, c, k, r, s, stride, pad, dil = geom
    return (pass_id == 1 and ops.CONV_MODE == 'h2' and ops.FUSE and ops.WINOGRAD_DGRAD and stride == 1 and pad == dil and
            ops._wino_eligible(k, c, r, s) and ops._wino_dgrad_eligible(k))


def time_dominant_kernel(dev, cfg, iters=10):
    """HIP-event timing of the dominant kernel of the step (CONFIGS[..]['dominant']; for configs[1] the DATA GRADIENT of
    decoder.conv_last.0, 3x3 4096->512 @64x64, N=2: 309.24 GFLOP, 12.6 % of the step's FLOPs) through the C ABI on pre-split
    operands, on the stream it is launched on (torch's current stream).  Returns (seconds per pass, GFLOP per pass, extras).
    Direct form: ONE launch (igemm_dma_kernel<SchH2,...>, tuned plan).  Winograd form (dominant_is_winograd): the pass is three
    launches -- input transform from the planes of dy, the batched GEMM over the 16 frequencies (the dominant launch), output
    transform -- and the time returned is that of ALL THREE (the algorithmic FLOPs of the layer are what the three produce
    together); the GEMM launch alone is timed as well (extras)."""
    import ctypes
    from mit_semseg import ops, _native, tuner
    L = _native.lib()
    vp = ctypes.c_void_p
    pass_id, geom, _ = cfg['dominant']
    n, h, w, c, k, r, s, stride, pad, dil = geom
    gflop = 2.0 * n * h * w * c * k * r * s * 1e-9
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    P = lambda t: vp(t.data_ptr())                              # noqa: E731

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    if dominant_is_winograd(cfg):
        wparam = torch.nn.Parameter((torch.randn(k, r, s, c, device=dev) * 0.01).permute(0, 3, 1, 2))      # KRSC memory
        ops.prepare_conv_weights([wparam])
        ut = ops.weight_wino_t(wparam)
        dy = torch.randn(n, h, w, k, device=dev) * 1e-3
        dyp = ops.SCHEMES['h2'].split(dy, n * h * w, k, k)
        t_pass = timed(lambda: ops._winograd_dgrad(L, dyp, ut, geom))          # the form the step runs (tuner.choose)
        tiles = L.semseg_winograd_tiles(n, h, w, dil)
        v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, k), dtype=torch.uint8, device=dev)
        _native.check(L.semseg_winograd_input_planes_h2(P(dyp), P(v), n, h, w, k, dil, st()), 'winograd_input_planes_h2')
        choice = tuner.tuned_plans().get(('h2', 4, tiles, n, dil, c, k, 3, 3, 1, h, w))
        form = int(choice[0]) if choice else 0
        executed = 3 * 16 * 2.0 * tiles * k * c * 1e-9
        if form >= 1:
            # GEMM over the 16 frequencies + output transform in ONE launch (wino_fused_kernel): the dominant launch
            dx = ops.empty_nhwc(n, c, h, w, dev)
            t_gemm = timed(lambda: _native.check(L.semseg_winograd_gemm_output_h2(P(v), P(ut), P(dx), c, n, h, w, k, c, dil, form - 1,
                                                                                 st()), 'winograd_gemm_output_h2'))
            return t_pass, gflop, {'form': 'winograd_fused', 'gemm_s': t_gemm, 'gemm_tile': form - 1, 'executed_gflop': executed}
        m = torch.empty((16 * tiles, c), dtype=torch.float32, device=dev)
        t_gemm = timed(lambda: _native.check(L.semseg_winograd_gemm_h2(P(v), P(ut), P(m), tiles, k, c, st()), 'winograd_gemm_h2'))
        plan = tuner.tuned_plans().get(('h2', 3, tiles, 1, 1, k, c, 3, 3, 1, 1, 1))
        return t_pass, gflop, {'form': 'winograd', 'gemm_s': t_gemm, 'gemm_tile': plan[0] if plan else None,
                               'executed_gflop': executed}
    if pass_id == 1:
        a = torch.randn(n, h, w, k, device=dev) * 1e-3                 # dy
        wt = torch.randn(c, r, s, k, device=dev) * 0.01                # CRSK
        out = torch.empty(n, h, w, c, device=dev)
        a_ch, w_rows, w_ch, out_ld = k, c * r * s, k, c
    else:
        a = torch.randn(n, h, w, c, device=dev)
        wt = torch.randn(k, r, s, c, device=dev) * 0.01                # KRSC
        out = torch.empty(n, h, w, k, device=dev)
        a_ch, w_rows, w_ch, out_ld = c, k * r * s, c, k
    if ops.CONV_MODE in ops.SCHEMES:
        sch = ops.SCHEMES[ops.CONV_MODE]
        ap, wp = sch.split(a, n * h * w, a_ch, a_ch), sch.split(wt, w_rows, w_ch, w_ch)

        def launch():
            ws = ops.workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
            if pass_id == 1:
                _native.check(sch.fn(L, 'dgrad')(P(ap), P(wp), P(out), out_ld, *geom, P(ws), ws.numel(), st()), 'dgrad_split')
            else:
                _native.check(sch.fn(L, 'fwd')(P(ap), P(wp), vp(0), P(out), out_ld, *geom, P(ws), ws.numel(), st()), 'fwd_split')
        tuner.ensure(ops.CONV_MODE, pass_id, geom, launch)
    else:
        def launch():
            ws = ops.workspace(L.semseg_conv2d_workspace_bytes(*geom), dev)
            if pass_id == 1:
                _native.check(L.semseg_conv2d_dgrad(P(a), k, P(wt), P(out), c, *geom, P(ws), ws.numel(), st()), 'dgrad')
            else:
                _native.check(L.semseg_conv2d_fwd(P(a), c, P(wt), vp(0), P(out), k, *geom, P(ws), ws.numel(), st()), 'fwd')
    return timed(launch), gflop, {'form': 'direct'}


def roofline_entry(kt, gflop, cfg, cfg_id, extras=None):
    """bound = MFMA.  `achieved` is ALGORITHMIC conv TFLOP/s (2*MACs of the layer / time of the pass)."""
    from mit_semseg import ops, tuner
    extras = extras or {'form': 'direct'}
    achieved = gflop / kt * 1e-3
    pass_id, geom, layer = cfg['dominant']
    what_pass = 'data gradient' if pass_id == 1 else 'forward'
    wino = extras.get('form') in ('winograd', 'winograd_fused')
    fused = extras.get('form') == 'winograd_fused'
    plan = (extras.get('gemm_tile'), 1) if wino else tuner.tuned_plans().get((ops.CONV_MODE, pass_id) + tuple(geom))
    pmc = pmc_lookup(ops.CONV_MODE, pass_id, geom, plan, extras.get('form') if wino else 'direct')
    if pmc is None:
        print('[bench] no PMC pass in profiles/pmc_index.json for the %s form of %s pass %d %s plan %s: roofline.traffic is null'
              % (extras.get('form'), ops.CONV_MODE, pass_id, list(geom), list(plan[:2]) if plan else None),
              file=sys.stderr, flush=True)
    traffic = (2 * pmc['fetch_kib'] + pmc['write_kib']) * 1024 if pmc else None
    alg_bytes = algorithmic_bytes(pass_id, geom, ops.CONV_MODE)
    if ops.CONV_MODE in SPLIT_TERMS:
        # the kernel issues 16-bit MFMAs (dense peak 2.5 PF); each fp32-accurate MAC costs `terms` 16-bit MACs, so the
        # path's own ceiling is 2500/terms algorithmic TFLOP/s and its MFMA-pipe utilisation is terms*achieved/2500
        terms, inst, what = SPLIT_TERMS[ops.CONV_MODE]
        clock = pmc.get('clock_ghz') if pmc else None
        if wino:
            # Winograd F(2x2,3x3): the GEMM launch executes 16/36 of the direct MACs (x `terms` products each)
            executed = extras['executed_gflop'] / extras['gemm_s'] * 1e-3
            if fused:
                kernel = ('Winograd-domain %s of %s: input transform from the h2 planes of dy + ONE wino_fused_kernel launch -- the GEMM '
                          'over the 16 frequencies AND the output transform: every block owns 128 tiles x 128 channels for all '
                          'frequencies, one accumulator set for M[f], four for the 2x2 outputs, ring form %s (%s; the dominant '
                          'launch: %.3f ms, %.1f executed 16-bit TFLOP/s); achieved = the layer\'s %.2f algorithmic GFLOP / the '
                          '%.3f ms of both launches, HIP events; traffic = FETCH_SIZE*2+WRITE_SIZE summed over the two launches '
                          '(PMC pass named in pmc_source): no fp32 intermediate exists'
                          % (what_pass, layer, extras.get('gemm_tile'), inst, extras['gemm_s'] * 1e3, executed, gflop, kt * 1e3))
            else:
                kernel = ('Winograd-domain %s of %s: input transform from the h2 planes of dy + ONE batched igemm_dma_kernel launch '
                          'over the 16 frequencies (%s, tile id %s; the dominant launch: %.3f ms, %.1f executed 16-bit TFLOP/s) + '
                          'output transform; achieved = the layer\'s %.2f algorithmic GFLOP / the %.3f ms of all three launches, HIP '
                          'events; traffic = FETCH_SIZE*2+WRITE_SIZE summed over the three launches (PMC pass named in pmc_source): '
                          'the fp32 intermediate of the 16 frequencies is written and read back'
                          % (what_pass, layer, inst, extras.get('gemm_tile'), extras['gemm_s'] * 1e3, executed, gflop, kt * 1e3))
        else:
            executed = terms * achieved
            kernel = ('split-%s implicit-GEMM %s (%s per fp32-accurate MAC block, %s), %s (%.2f GFLOP/launch algorithmic, '
                      '%.3f ms/launch, HIP events; traffic = FETCH_SIZE*2+WRITE_SIZE of the PMC pass named in pmc_source, '
                      'bytes/launch)' % (ops.CONV_MODE, what_pass, what, inst, layer, gflop, kt * 1e3))
        return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), 'traffic': traffic,
                'form': extras.get('form'),
                'executed_16bit_mfma_tflops': round(executed, 1),
                'mfma_pipe_utilisation': round(executed / PEAK_BF16_MFMA_TFLOPS, 4),
                'path_ceiling_tflops': round(PEAK_BF16_MFMA_TFLOPS / terms * (2.25 if wino else 1.0), 1),
                'frac_of_path_ceiling': round(achieved / (PEAK_BF16_MFMA_TFLOPS / terms * (2.25 if wino else 1.0)), 4),
                'frac_of_fp32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                'algorithmic_bytes': alg_bytes,
                'clock_ghz_under_load': clock,
                'mfma_pipe_utilisation_at_measured_clock': round(executed / (PEAK_BF16_MFMA_TFLOPS * clock / 2.4), 4)
                if clock else None,
                'pmc_source': pmc.get('source') if pmc else None,
                'plan_tile_split': list(plan[:2]) if plan else None,
                'dominant_launch_ms': round((extras['gemm_s'] if wino else kt) * 1e3, 4),
                'kernel': kernel}
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic, 'algorithmic_bytes': alg_bytes,
            'pmc_source': pmc.get('source') if pmc else None,
            'kernel': 'exact-fp32 MFMA implicit-GEMM %s, %s (%.2f GFLOP/launch, %.3f ms/launch, HIP events)'
                      % (what_pass, layer, gflop, kt * 1e3)}


def cpu_baseline(cfg, steps=3):
    """the reference's CPU path timed on this host in the same run -- oracle/cpu_baseline.py in a subprocess (the
    UNMODIFIED reference where /root/reference is importable, else the oracle port over the same torch CPU kernels): 1 warm-up
    + `steps` timed training steps of the same 2x512x512 synthetic batch.  torch's CPU conv/BN kernels stop scaling (and
    thrash) far below the 256 hardware threads of the GPU box (203 s per step at 256 threads), so at most 32 threads are used;
    `cores` = threads used, `host_cores` = what the host has."""
    import subprocess
    host = os.cpu_count() or 1
    threads = min(host, int(os.environ.get('SEMSEG_CPU_BASELINE_THREADS', '32')))
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_baseline.py'), '--config', cfg['name'], '--n', '2', '--h', '512',
           '--w', '512', '--threads', str(threads), '--steps', str(steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
        out = json.loads(line)
    except Exception as e:                                    # never lose the GPU line to the CPU leg
        return {'value': None, 'unit': 'images/sec', 'cores': threads, 'kind': 'port', 'sample': 'failed: %r' % (e,)}
    # BASELINE.md 3 says torch.set_num_threads(os.cpu_count()): that run too, so the recipe is visibly followed -- one timed step after
    # one warm-up step, under a time limit (a 256-thread host needs minutes per step: the line then says so instead of a number)
    if host > threads:
        limit = float(os.environ.get('SEMSEG_CPU_BASELINE_ALL_CORES_S', '60'))
        t0 = time.perf_counter()
        try:
            cmd2 = [a for a in cmd]
            cmd2[cmd2.index('--threads') + 1] = str(host)
            cmd2[cmd2.index('--steps') + 1] = '1'
            r = subprocess.run(cmd2, capture_output=True, text=True, timeout=limit)
            l2 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
            out['all_cores'] = {'cores': host, 'value': l2['value'], 'unit': l2['unit'], 'sample': l2.get('sample')}
        except subprocess.TimeoutExpired:
            out['all_cores'] = {'cores': host, 'value': None,
                                'sample': 'did not finish 1 warm-up + 1 timed step within %.0f s (torch CPU convolutions thrash at this '
                                          'thread count: 203 s per step measured at 256 threads in round 2)' % limit}
        except Exception as e:
            out['all_cores'] = {'cores': host, 'value': None, 'sample': 'failed: %r' % (e,)}
        out['all_cores']['wall_s'] = round(time.perf_counter() - t0, 1)
    return out


STAGES = ('comm', 'peer', 'segmented', 'graph')


def ddp_graph_selftest(timeout_s=240):
    """world > 1: decide what the real run may rely on.  Every rank runs tools/probes/ddp_graph_selftest.py (a small model through
    exactly the data-parallel code paths) in a CHILD process, on its own rendezvous port, BEFORE this process touches the GPU or
    RCCL; a child that fails or does not finish in time is killed.  Returns the stages this rank's child reached:
      'comm'       the C ABI's own RCCL communicator came up and summed correctly (else the real run sets SEMSEG_NATIVE_COMM=0
                   and the collectives go through torch.distributed),
      'peer'       the one-node peer exchange of the SyncBN payloads (csrc/peer.hip: IPC-mapped inboxes, xGMI peer stores) came
                   up on every rank and summed correctly (else SEMSEG_PEER=0: RCCL / torch.distributed carry them),
      'segmented'  the segmented hipGraph executor trained (else SEMSEG_DDP_SEGMENTED=0: eager launches),
      'graph'      the whole step, RCCL included, replayed as ONE hipGraph (then SEMSEG_DDP_GRAPH=1).
    The ranks agree on the outcome afterwards (minimum over ranks)."""
    import subprocess
    import tempfile
    env = dict(os.environ)
    env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + 17)
    env.pop('TORCHELASTIC_USE_AGENT_STORE', None)       # rank 0 of the children opens its own store on the new port
    env.pop('SEMSEG_TUNE_CACHE', None)
    env['SEMSEG_TUNE'] = '0'                 # the library's heuristic launch plans: the child checks mechanisms, not speed
    env.setdefault('SEMSEG_PEER', '1')       # the peer exchange is opt-in (comm.peer_enabled): the child is where it earns its place
    script = os.path.join(ROOT, 'tools', 'probes', 'ddp_graph_selftest.py')
    out, rc, why = '', None, 'not started'
    with tempfile.TemporaryFile() as log:
        try:
            p = subprocess.Popen([sys.executable, script], env=env, stdout=log, stderr=subprocess.STDOUT)
            try:
                rc = p.wait(timeout=timeout_s)
                why = 'exit code %d' % rc
            except subprocess.TimeoutExpired:
                p.kill()                                 # the exact child started above
                p.wait()
                why = 'no result within %d s (killed)' % timeout_s
        except OSError as e:
            why = str(e)
        log.seek(0)
        out = log.read().decode('utf-8', 'replace')
    # every rank's child prints the markers of the stages it passed into its own log (a child that dies in a later stage -- the
    # whole-step graph is the risky one -- keeps the earlier ones); the all-reduce(min) in main() combines the ranks
    stages = {'comm': 'COMM_OK' in out, 'peer': 'PEER_OK' in out, 'segmented': 'SEGMENTED_OK' in out,
              'graph': rc == 0 and 'GRAPH_OK' in out}
    if rc != 0 and env.get('RANK', '0') == '0':
        print('[bench] data-parallel self-test: %s; stages reached %s\n%s' % (why, stages, out[-1500:]), file=sys.stderr, flush=True)
    return stages


def collectives_used(world):
    """which transport carried what, with the transports' own count of the ranks: `rccl_ranks` = ncclCommCount of the two native
    communicators (C ABI semseg_comm_count), `peer_world` = ranks attached to the xGMI peer exchange"""
    if world == 1:
        return None
    from mit_semseg import comm
    buckets = 'C ABI RCCL communicator (semseg_comm_*)' if comm.active() else 'torch.distributed'
    syncbn = 'xGMI peer exchange kernel (semseg_peer_*)' if comm.peer_active() else buckets
    import torch.distributed as dist
    return {'syncbn': syncbn, 'gradient_buckets': buckets, 'rccl_ranks': comm.rccl_ranks(), 'peer_world': comm.peer_world(),
            'torch_distributed_backend': dist.get_backend(), 'torch_distributed_world': dist.get_world_size()}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here -- `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` (what the
    reference does with one command, train.py:184-190; drivers.launch does the same for train.py) -- and pass rank 0's JSON line
    through.  Refuses (exit code 2, no JSON line) when the node has fewer GPUs than ranks, unless SEMSEG_BENCH_DEVICE names the
    one GPU that all ranks are meant to share (functional check of the N > 1 path on a 1-GPU box, gloo transport)."""
    import socket
    import subprocess
    n = args.gpus
    probe = os.environ.get('SEMSEG_BENCH_LAUNCH_PROBE') == '1'
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    if not probe and have < n:
        if 'SEMSEG_BENCH_DEVICE' not in env:
            print('[bench] --gpus %d but this node has %d GPU(s): refusing to run (and to print a line with n_gpus != --gpus). '
                  'SEMSEG_BENCH_DEVICE=<index> runs all ranks on ONE GPU over gloo as a functional check.' % (n, have),
                  file=sys.stderr, flush=True)
            return 2
        env.setdefault('SEMSEG_DIST_BACKEND', 'gloo')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env['SEMSEG_BENCH_SELF_LAUNCHED'] = '1'
    return subprocess.call(cmd, env=env)


def launch_probe(args):
    """SEMSEG_BENCH_LAUNCH_PROBE=1: every rank joins the process group (gloo), the ranks count themselves, rank 0 prints the line's
    launch-related keys and everybody leaves -- the launch plumbing without a GPU (tests/test_distributed_cpu.py)."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        counted = int(t.item())
    else:
        counted = 1
    if rank == 0:
        print(json.dumps({'probe': True, 'n_gpus': world, 'ranks_counted': counted, 'gpus_arg': args.gpus,
                          'self_launched': os.environ.get('SEMSEG_BENCH_SELF_LAUNCHED') == '1'}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def other_configs(skip, steps, warmup, budget_s=100):
    """BASELINE configs[2..4] measured next to the headline (outside its timed region, one after the other, nothing else running
    on the host or the GPU): one `bench.py --config N` leg each, in a child process (own allocator pool, own launch plans).
    Every leg is labelled with what it ran: mean pixels per image, distinct shapes, train GFLOP per image.  Returns {cfg: summary}."""
    import subprocess
    out = {}
    for cid in sorted(CONFIGS):
        if cid == skip:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), '--config', str(cid), '--gpus', '1', '--steps', str(steps), '--warmup',
               str(warmup), '--no-cpu-baseline', '--no-other-configs', '--repeats', '0', '--no-box', '--no-scaling-model']
        limit = budget_s
        if CONFIGS[cid].get('variable'):
            # BASELINE's rule (dataset.py:110-142 over the ADE20K size histogram) on the first 16 distinct shapes of the stream,
            # each seen twice (eager + capture) before the timed steps; K steps cycle through the stream's own repetition pattern
            cmd += ['--shapes', '16', '--steps', str(max(steps, 32))]
            limit = budget_s + 80
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit)
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
            c = line['config']
            out[str(cid)] = {'workload': CONFIGS[cid]['yaml'], 'img_s': line['value'], 'ms_per_step': line['ms_per_step'],
                             'launch': c['launch'], 'step_conv_tflops': c['step_conv_tflops_per_gpu'],
                             'mean_px_per_image': c['mean_px_per_image'], 'mean_side_px': round(c['mean_px_per_image'] ** 0.5, 1),
                             'distinct_shapes_timed': c['distinct_shapes_timed'],
                             'train_gflop_per_image': c['train_gflop_per_image'],
                             'roofline_frac': line['roofline']['frac'], 'roofline_achieved_tflops': line['roofline']['achieved'],
                             'final_loss': c['final_loss'], 'steps': line['steps'],
                             'leg_wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:                                    # never lose the headline to a side leg
            out[str(cid)] = {'workload': CONFIGS[cid]['yaml'], 'img_s': None, 'error': repr(e)[:200],
                             'leg_wall_s': round(time.perf_counter() - t0, 1)}
        if CONFIGS[cid].get('variable') and out[str(cid)].get('img_s'):
            # ... and the RAW stream of the same rule beside its steady state (round-4 review): 100 untimed + 300 timed steps in
            # stream order, the recording passes of new shapes INSIDE the timed region -- the first minutes of
            # a training run, before the ~470 shapes of the ADE20K list have all been captured (tools/shape_stream_sim.py)
            raw = [sys.executable, os.path.abspath(__file__), '--config', str(cid), '--gpus', '1', '--shapes', '0', '--steps', '300',
                   '--warmup', '100', '--no-cpu-baseline', '--no-other-configs', '--repeats', '0', '--no-box', '--no-scaling-model']
            t0 = time.perf_counter()
            try:
                r = subprocess.run(raw, capture_output=True, text=True, timeout=budget_s + 140)
                line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
                c = line['config']
                ev = c['shape_events_timed']
                out[str(cid)]['raw_stream'] = {
                    'img_s': line['value'], 'ms_per_step': line['ms_per_step'], 'steps': line['steps'], 'warmup': line['warmup'],
                    'fraction_of_steady_state': round(line['value'] / out[str(cid)]['img_s'], 3),
                    'mean_px_per_image': c['mean_px_per_image'], 'distinct_shapes_timed': c['distinct_shapes_timed'],
                    'events_timed': ev,
                    'host_ms_per_capture': round(1e3 * ev['capture_host_s'] / max(1, ev['captured']), 1),
                    'of_which_instantiate_ms': round(1e3 * ev.get('instantiate_host_s', 0.0) / max(1, ev['captured']), 1),
                    'host_ms_per_first_sight': round(1e3 * ev['eager_host_s'] / max(1, ev['eager']), 1),
                    'note': 'stream order, graph LRU 512; a new shape is recorded into its hipGraph the first time it is seen (one Python pass, '
                            'no device synchronize) and replayed from then on; an eager pass only where a launch plan had to be timed',
                    'leg_wall_s': round(time.perf_counter() - t0, 1)}
            except Exception as e:
                out[str(cid)]['raw_stream'] = {'img_s': None, 'error': repr(e)[:200], 'leg_wall_s': round(time.perf_counter() - t0, 1)}
    return out


def box_probe(dev):
    """What THIS box sustains, measured in this process right after the timed steps (C ABI semseg_probe_*, csrc/probe.hip): dense
    fp16 MFMA rate and the shader clock held under it, float4 copy bandwidth, per-node cost of a hipGraph chain of empty kernels.
    The same tree measured 137 ... 153 img/s on different boxes in round 3; with this block a slow box reads as a slow box."""
    import ctypes
    from mit_semseg import _native
    L = _native.lib()
    vp = ctypes.c_void_p
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return ts
    try:
        blocks, iters = 1024, 12000
        sink = torch.zeros(4, device=dev)
        cyc = torch.zeros(blocks, dtype=torch.int64, device=dev)
        ts = timed(lambda: _native.check(L.semseg_probe_mfma_f16(vp(sink.data_ptr()), blocks, iters, vp(cyc.data_ptr()), st()),
                                         'probe_mfma'), 4)
        flop = blocks * 4.0 * iters * 8 * 2 * 32 * 32 * 16
        t = sorted(ts)[len(ts) // 2]
        out['mfma_f16_dense_tflops'] = round(flop / t * 1e-12, 1)
        out['mfma_f16_frac_of_2500'] = round(flop / t * 1e-12 / PEAK_BF16_MFMA_TFLOPS, 4)
        # every SIMD issues one 32x32x16 MFMA per 32 cycles at best: the rate IS a lower bound of the clock held by this
        # (memory-free, low-power) loop; the dense GEMM kernels of the step hold less (roofline.clock_ghz_under_load, PMC)
        out['mfma_implied_clock_ghz'] = round(flop / t * 1e-12 / PEAK_BF16_MFMA_TFLOPS * 2.4, 3)
        out['s_memtime_ticks_per_ns'] = round(float(cyc.double().mean().item()) / ts[-1] * 1e-9, 3)
        out['mfma_probe'] = '%d blocks x 4 waves x %d x 8 v_mfma_f32_32x32x16_f16, %.2f ms, median of %d' % (blocks, iters, t * 1e3, len(ts))
    except Exception as e:
        out['mfma_error'] = repr(e)[:160]
    try:
        nbytes = 1 << 30
        a = torch.empty(nbytes // 4, device=dev).normal_()
        b = torch.empty_like(a)
        ts = timed(lambda: _native.check(L.semseg_probe_copy(vp(a.data_ptr()), vp(b.data_ptr()), nbytes, st()), 'probe_copy'), 5)
        t = sorted(ts)[len(ts) // 2]
        out['hbm_copy_GBps'] = round(2.0 * nbytes / t * 1e-9, 1)
        out['hbm_copy_frac_of_8000'] = round(2.0 * nbytes / t * 1e-9 / 8000.0, 4)
        ts = timed(lambda: b.copy_(a), 5)                     # cross-check: the runtime's own device-to-device copy
        out['hbm_copy_runtime_GBps'] = round(2.0 * nbytes / sorted(ts)[len(ts) // 2] * 1e-9, 1)
        out['hbm_probe'] = '1 GiB read + 1 GiB written per launch (16 B per lane, 4 loads in flight), median of 5; bytes counted both ways'
        del a, b
    except Exception as e:
        out['copy_error'] = repr(e)[:160]
    try:
        nodes = 2000
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(nodes):
                L.semseg_probe_empty(st())
        ts = timed(g.replay, 5)
        out['graph_node_us'] = round(sorted(ts)[len(ts) // 2] / nodes * 1e6, 3)
        out['graph_probe'] = 'replay of a captured chain of %d empty one-wave kernels' % nodes
        del g
    except Exception as e:
        out['graph_error'] = repr(e)[:160]
    try:
        out['device'] = torch.cuda.get_device_name(dev)
        out['cus'] = torch.cuda.get_device_properties(dev).multi_processor_count
    except Exception:
        pass
    return out


def scaling_model_block(step, sm, feed, t1_ms):
    """no multi-GPU box -> measured 1-GPU step + analytic comm model.  The step is re-captured with timestamp
    markers (csrc/probe.hip) at forward end, at the completion of every gradient bucket of parallel.plan_bucket_groups, at
    backward end and at step end; three replays, the last one's timeline scaled to the measured step feeds
    mit_semseg/scaling_model.py.  MODEL, NOT MEASURED."""
    from mit_semseg import scaling_model as smod
    enc, dec = sm.encoder, sm.decoder
    probe = smod.TimelineProbe(list(enc.parameters()) + list(dec.parameters()))
    # the timeline of a RANK: under gradient buckets the deferred / batched weight gradients are finished by the hook that completes
    # a bucket, before its all-reduce (parallel.GradientBuckets._launch); the probe's bucket hooks flush the same way, so the markers
    # are captured -- and the step is timed -- in the form a rank runs it (one flush per bucket instead of one after backward)
    t_rank_ms = None
    try:
        step.timeline = probe
        step._graphs.clear()
        for _ in range(4):                       # capture + three replays with the markers inside the graph
            step.step(feed)
        torch.cuda.synchronize()
        ticks = probe.read()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step.step(feed)
        e1.record()
        torch.cuda.synchronize()
        t_rank_ms = e0.elapsed_time(e1) / reps
    finally:
        step.timeline = None
        step._graphs.clear()
        probe.detach()
    return smod.model_line(t1_ms, ticks, probe.bucket_bytes, smod.syncbn_payloads(sm), t_rank_ms=t_rank_ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='N = 1, default config: skip the short legs of BASELINE configs[2..4] (config.other_configs)')
    ap.add_argument('--repeats', type=int, default=3,
                    help='after the contract window (W warm-up + exactly K timed steps = `value`): this many more timed windows of '
                         'max(K, 50) steps each; their ms/step, median, min and max go into config.repeat_windows')
    ap.add_argument('--no-box', action='store_true', help='skip the box probes (MFMA rate, HBM copy, graph node latency)')
    ap.add_argument('--no-scaling-model', action='store_true', help='N = 1: skip the analytic 2/4/8-GPU model')
    ap.add_argument('--shapes', type=int, default=16,
                    help='config 3: number of distinct batch shapes of the stream that the run cycles through, all seen (and '
                         'captured) before the timed steps; 0 = the raw stream, cold path included')
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS),
                    help='index into BASELINE.json configs (default 1: the configuration the metric is quoted on)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    # --gpus is the number of ranks of THIS job: with no launcher around us (WORLD_SIZE unset) start them here; under a
    # launcher the two must agree -- a line whose n_gpus differs from --gpus is never printed
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    if int(env_world or '1') != args.gpus:
        print('[bench] --gpus %d but the launcher started WORLD_SIZE=%s rank(s): refusing to run' % (args.gpus, env_world),
              file=sys.stderr, flush=True)
        sys.exit(2)
    if os.environ.get('SEMSEG_BENCH_LAUNCH_PROBE') == '1':
        sys.exit(launch_probe(args))

    import __graft_entry__ as ge
    ge.build()
    from mit_semseg.parallel import init_distributed, NativeDataParallel
    from mit_semseg.engine import TrainStep
    import torch.distributed as dist

    # Data-parallel runs launch eagerly unless SEMSEG_DDP_GRAPH says otherwise.  With nothing said, let a self-test decide:
    # if the captured RCCL all-reduces work for a small model on THIS node (every rank, inside a time limit), the real run
    # replays a hipGraph as the single-GPU run does; if not, nothing changes.
    selftest_ok = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and 'SEMSEG_DDP_GRAPH' not in os.environ and not args.no_graph and \
            not os.environ.get('SEMSEG_DIST_BACKEND') and not os.environ.get('SEMSEG_BENCH_DEVICE'):
        selftest_ok = ddp_graph_selftest()
        if not selftest_ok['comm']:
            os.environ['SEMSEG_NATIVE_COMM'] = '0'      # before NativeDataParallel is built: torch.distributed carries the collectives
    rank, world, local = init_distributed()
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    if os.environ.get('SEMSEG_BENCH_DEVICE'):          # several ranks on one GPU (gloo): functional check of the N>1 path
        local = int(os.environ['SEMSEG_BENCH_DEVICE'])
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if selftest_ok is not None:
        flag = torch.tensor([int(selftest_ok[k]) for k in STAGES], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # every rank's child must have reached the stage
        selftest_ok = dict(zip(STAGES, (bool(v) for v in flag.tolist())))
        if not selftest_ok['comm']:
            os.environ['SEMSEG_NATIVE_COMM'] = '0'
        # the peer exchange (opt-in) is switched on only when EVERY rank's child trained the small model through it -- fused BN
        # kernels exchanging inside hipGraph segments, bucket all-reduces on the side stream between them, bit-identical replicas
        if os.environ.get('SEMSEG_PEER', '1') != '0':
            os.environ['SEMSEG_PEER'] = '1' if (selftest_ok['peer'] and selftest_ok['segmented']) else '0'
        if not selftest_ok['segmented']:
            os.environ['SEMSEG_DDP_SEGMENTED'] = '0'
        # One hipGraph for the whole step needs the gradient buckets' side stream INSIDE the graph, and a graph with parallel
        # chains is submitted node by node (profiles/r3i-m_ab_tile_forms.txt: +3.4 ms on this step).  With the peer exchange up
        # the segmented executor has only the buckets between its (linear) segments, so it is the better mode (the child does not
        # even try the single graph then); without it the alternative is 122 host-issued collectives per step, and the single
        # graph is preferred when it works.
        if selftest_ok['graph'] and not (selftest_ok['peer'] and selftest_ok['segmented']):
            os.environ['SEMSEG_DDP_GRAPH'] = '1'
    sm = build_model(dev, cfg)
    if world > 1:
        NativeDataParallel(sm)          # enables SyncBN statistics all-reduce over RCCL
    if cfg.get('variable'):
        shapes, pool = variable_feeds(dev, rank, cfg, args.warmup + args.steps, distinct=args.shapes)
        feeds = [pool[hw] for hw in shapes]
    else:
        shapes = [(512, 512)] * (args.warmup + args.steps)
        feeds = [synth_feed(dev, rank, cfg)] * (args.warmup + args.steps)
    step = TrainStep(sm, lr_encoder=0.02, lr_decoder=0.02, max_iters=5000 * 20,
                     graph=not args.no_graph)      # world > 1: eager unless SEMSEG_DDP_GRAPH=1 (TrainStep)

    if cfg.get('variable') and args.shapes:
        # steady state of the variable-size path: every shape of the run has been seen and recorded into its hipGraph (TrainStep: at its
        # first sight after the two eager warm-up steps) before the warm-up / timed steps, as in a long training run; `--shapes 0`
        # times the cold path instead (the recording passes of new shapes inside the timed region)
        for _ in range(2):
            for hw in pool:
                step.step(pool[hw])
    for i in range(args.warmup):
        loss, acc = step.step(feeds[i])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    before = dict(step.stats)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss, acc = step.step(feeds[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    imgs = 2 * world * args.steps
    value = imgs / dt
    lossv = loss.item()
    timed = {k: step.stats[k] - before[k] for k in step.stats}          # launch mode of the contract window
    # more windows of the same replayed step, so that the one contract number can be read against its own spread
    # (round-3 review: one 0.29 s sample; driver 137.2 vs builder boxes 147-153 with nothing in the line to tell why)
    windows = []
    if args.repeats > 0:
        n_rep = max(args.steps, 50)
        for _ in range(args.repeats):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            for i in range(n_rep):
                step.step(feeds[args.warmup + i % args.steps])
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            rt = time.perf_counter() - r0
            if world > 1:
                t = torch.tensor([rt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rt = t.item()
            windows.append(rt / n_rep * 1e3)

    px = [h * w for h, w in shapes[args.warmup:]]
    gflop_img = cfg['gflop'] * (sum(px) / len(px)) / (512.0 * 512.0)
    if rank == 0:
        try:
            kt, kgflop, kextras = time_dominant_kernel(dev, cfg)
        except Exception as e:                     # never lose the line to the roofline leg: fall back to the direct kernel
            print('[bench] timing the dominant pass in its Winograd form failed (%r): timing the direct kernel instead' % (e,),
                  file=sys.stderr, flush=True)
            from mit_semseg import ops as _ops
            _ops.WINOGRAD_DGRAD = False
            kt, kgflop, kextras = time_dominant_kernel(dev, cfg)
        per_gpu = value / world
        out = {
            'metric': 'train images/sec (whole job) @%s bs2/GPU' % ('multi-scale variable-size' if cfg.get('variable') else '512x512'), 'value': round(value, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[ops_mode()], 'data': 'synthetic',
            'config': {'workload': '%s (BASELINE configs[%d]): full train step (fwd+NLL loss+bwd+2xSGD), bs 2/GPU, %s, 150 '
                                   'classes, labels at 1/%d' % (
                                       cfg['yaml'], args.config,
                                       ('variable-size batches (dataset.py:110-142 rule over the ADE20K size histogram): mean %.0f '
                                        'px/img = %.0f^2, %d distinct shapes in the timed steps%s'
                                        % (sum(px) / len(px), (sum(px) / len(px)) ** 0.5, len(set(shapes[args.warmup:])),
                                           ', all seen before the timed region (steady state)' if args.shapes else ' (cold path '
                                           'included: the recording passes of new shapes are timed)'))
                                       if cfg.get('variable') else '512x512x3', cfg['rate']),
                       'global_batch': 2 * world, 'parallelism': 'dp%d' % world,
                       'launch': (('segmented hipGraphs (%s)' % step._graph.counts() if type(step._graph).__name__ == 'SegmentedStep'
                                   else 'hipGraph replay') if timed['replayed'] == args.steps else
                                  'eager' if timed['replayed'] == 0 else
                                  'per-shape hipGraphs: %(replayed)d of the timed steps replayed (%(captured)d captured in the '
                                  'timed region), %(eager)d eager passes' % timed),
                       'ddp_graph_selftest': selftest_ok,
                       'collectives': collectives_used(world),
                       'conv_path': ops_mode(),
                       'side_by_side': side_by_side_stats(),
                       'launch_plans': launch_plan_provenance(),
                       'images_per_sec_per_gpu': round(per_gpu, 3),
                       'step_conv_tflops_per_gpu': round(per_gpu * gflop_img * 1e-3, 2),
                       'train_gflop_per_image': round(gflop_img, 1),
                       'mean_px_per_image': round(sum(px) / len(px), 1),
                       'distinct_shapes_timed': len(set(shapes[args.warmup:])),
                       'shape_events_timed': {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timed.items()},
                       'repeat_windows': ({'steps_per_window': max(args.steps, 50),
                                           'ms_per_step': [round(w, 3) for w in windows],
                                           'median_ms': round(sorted(windows)[len(windows) // 2], 3),
                                           'min_ms': round(min(windows), 3), 'max_ms': round(max(windows), 3),
                                           'median_img_s': round(2e3 * world / sorted(windows)[len(windows) // 2], 2),
                                           'contract_window_ms': round(dt / args.steps * 1e3, 3),
                                           'note': '`value` / `ms_per_step` are the contract window (W warm-up + exactly K '
                                                   'steps); these windows follow it on the same replayed graph'}
                                          if windows else None),
                       'final_loss': round(lossv, 5)},
            'roofline': roofline_entry(kt, kgflop, cfg, args.config, kextras),
        }
        if world != args.gpus:                  # cannot happen past the checks at the top; never print a mislabelled line
            raise SystemExit('[bench] world %d != --gpus %d' % (world, args.gpus))
        if not args.no_box:
            out['box'] = box_probe(dev)
        if world == 1 and not args.no_scaling_model and not cfg.get('variable') and not args.no_graph:
            try:
                out['scaling_model'] = scaling_model_block(step, sm, feeds[0], dt / args.steps * 1e3)
            except Exception as e:                   # never lose the line to the model leg
                out['scaling_model'] = {'error': repr(e)[:300]}
        # the side legs run ONE AFTER THE OTHER (round-3 review / ADVICE: the 32-thread CPU leg used to run beside the child GPU
        # legs and both numbers moved): first the GPU legs of configs[2..4] on an otherwise idle host, then the CPU baseline
        if world == 1 and args.config == 1 and not args.no_other_configs:
            del step, sm, feeds                  # the legs run in child processes on the same GPU: hand the memory back first
            torch.cuda.empty_cache()
            out['config']['other_configs'] = other_configs(args.config, args.steps, args.warmup)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
            # the GPU box has no /root/reference, so `kind` is "port" there; what the port is worth against the UNMODIFIED reference was
            # measured where both run -- the build container, same thread count, interleaved (round-5 review, item 7b)
            try:
                pair = json.load(open(os.path.join(ROOT, 'profiles', 'r8_cpu_baseline_port_vs_reference.json')))
                out['cpu_baseline']['port_vs_reference'] = {'ratio': pair['port_vs_reference'], 'reference_img_s': pair['reference_img_s'],
                                                            'port_img_s': pair['port_img_s'], 'threads': 8,
                                                            'source': 'profiles/r8_cpu_baseline_port_vs_reference.json (build container)'}
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        from mit_semseg import comm
        dist.barrier()
        comm.peer_check()
        comm.peer_destroy()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
