"""Benchmark of the MI355X hot path on BASELINE.json's metric: training images/sec at 512x512, bs 2 per GPU,
ade20k-resnet50dilated-ppm_deepsup (configs[1]), synthetic ADE20K-shaped batches, seeded random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

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
TRAIN_GFLOP_PER_IMG = {'resnet50dilated+ppm_deepsup': 1224.2}
FWD_GFLOP_CONV_LAST = 309.24          # decoder.conv_last.0: 3x3 4096->512 @64x64, N=2 (per launch; fwd and dgrad alike)
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 per the guide's gfx950
# correction + WRITE_SIZE); measured offline with tools/gpu_pmc.sh, summaries under profiles/ (None = not measured yet)
F32_CONV_LAST_HBM_BYTES = 1.78e9      # profiles/r1b_pmc_conv_last_fwd_wgrad.txt: 873044 KB x 2 + 32768 KB
S3_CONV_LAST_HBM_BYTES = None


# profiles/r2s_pmc_conv_last_dgrad_h2.txt: FETCH_SIZE 168153 KiB x 2 + WRITE_SIZE 131072 KiB (the fp32 dx, no split-K);
# the algorithmic bytes of this launch are 226.5 MB (dy planes 16.8 MB + w planes 75.5 MB + 134.2 MB fp32 output)
H2_CONV_LAST_HBM_BYTES = (2 * 168153 + 131072) * 1024
H2_CONV_LAST_CLOCK_GHZ = 1.55     # SQ_WAVE_CYCLES x 4 / waves / duration of the same PMC pass: the MFMA-dense kernel runs
                                  # power-limited well below the 2.4 GHz the 2.5 PFLOP/s peak is quoted at
DTYPE = {'h2': 'f32 (fp32 in/out/accumulate; products on the fp16 MFMA via a scaled 2-way fp16 split, 3 terms, 2^-22 per product)',
         's3': 'f32 (fp32 in/out/accumulate; products on the bf16 MFMA via an exact 3-way bf16 split, 6 terms)', 'f32': 'f32'}
# MFMA products per fp32-accurate MAC block and the issued instruction, per split scheme
SPLIT_TERMS = {'h2': (3, 'v_mfma_f32_32x32x16_f16', '2-way fp16 split, 3 fp16 MFMAs'),
               's3': (6, 'v_mfma_f32_32x32x16_bf16', '3-way bf16 split, 6 bf16 MFMAs')}


def ops_mode():
    from mit_semseg import ops
    return ops.CONV_MODE


def build_model(dev, seed=304):
    from mit_semseg.models import ModelBuilder, SegmentationModule
    from mit_semseg.models import resnet
    from mit_semseg.models.models import ResnetDilated
    torch.manual_seed(seed)
    # random init exactly as the reference builds it without a checkpoint (resnet.py:118-124, models.py:52-61)
    enc = ResnetDilated(resnet.resnet50(pretrained=False), dilate_scale=8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=2048, num_class=150)
    crit = nn.NLLLoss(ignore_index=-1)
    return SegmentationModule(enc, dec, crit, deep_sup_scale=0.4).to(dev).train()


def synth_feed(dev, rank, n=2, h=512, w=512, seg_rate=8):
    g = torch.Generator().manual_seed(304 + rank)
    img = torch.randn(n, 3, h, w, generator=g)
    lab = torch.randint(-1, 150, (n, h // seg_rate, w // seg_rate), generator=g)
    return {'img_data': img.to(dev), 'seg_label': lab.to(dev)}


def time_dominant_kernel(dev, iters=10):
    """HIP-event timing of the dominant kernel of the step -- the implicit-GEMM DATA GRADIENT of decoder.conv_last.0 (3x3,
    4096->512 @64x64, N=2: 309.24 GFLOP, 12.6 % of the step's FLOPs; its forward runs through the Winograd path since r2m,
    so the largest single launch of the step is this one) -- through the C ABI on pre-split operands, i.e. ONLY the conv
    entry point (igemm_dma_kernel<SchH2,256,256> on the default h2 path, tuned plan: no split-K) on the stream it is
    launched on (torch's current stream)."""
    import ctypes
    from mit_semseg import ops, _native, tuner
    L = _native.lib()
    vp = ctypes.c_void_p
    n, h, w, c, k = 2, 64, 64, 4096, 512
    geom = (n, h, w, c, k, 3, 3, 1, 1, 1)
    dy = torch.randn(n, h, w, k, device=dev) * 1e-3
    wtt = torch.randn(c, 3, 3, k, device=dev) * 0.01               # CRSK
    dx = torch.empty(n, h, w, c, device=dev)
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    P = lambda t: vp(t.data_ptr())                              # noqa: E731
    if ops.CONV_MODE in ops.SCHEMES:
        sch = ops.SCHEMES[ops.CONV_MODE]
        dys, wts = sch.split(dy, n * h * w, k, k), sch.split(wtt, c * 9, k, k)

        def launch():
            ws = ops.workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
            _native.check(sch.fn(L, 'dgrad')(P(dys), P(wts), P(dx), c, *geom, P(ws), ws.numel(), st()), 'dgrad_split')
        tuner.ensure(ops.CONV_MODE, 1, geom, launch)
    else:
        def launch():
            ws = ops.workspace(L.semseg_conv2d_workspace_bytes(*geom), dev)
            _native.check(L.semseg_conv2d_dgrad(P(dy), k, P(wtt), P(dx), c, *geom, P(ws), ws.numel(), st()), 'dgrad')
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def roofline_entry(kt):
    """bound = MFMA.  `achieved` is ALGORITHMIC conv TFLOP/s (2*MACs of the layer / kernel time)."""
    from mit_semseg import ops
    achieved = FWD_GFLOP_CONV_LAST / kt * 1e-3
    if ops.CONV_MODE in SPLIT_TERMS:
        # the kernel issues 16-bit MFMAs (dense peak 2.5 PF); each fp32-accurate MAC costs `terms` 16-bit MACs, so the
        # path's own ceiling is 2500/terms algorithmic TFLOP/s and its MFMA-pipe utilisation is terms*achieved/2500
        terms, inst, what = SPLIT_TERMS[ops.CONV_MODE]
        traffic = H2_CONV_LAST_HBM_BYTES if ops.CONV_MODE == 'h2' else S3_CONV_LAST_HBM_BYTES
        return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), 'traffic': traffic,
                'executed_16bit_mfma_tflops': round(terms * achieved, 1),
                'mfma_pipe_utilisation': round(terms * achieved / PEAK_BF16_MFMA_TFLOPS, 4),
                'path_ceiling_tflops': round(PEAK_BF16_MFMA_TFLOPS / terms, 1),
                'frac_of_fp32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                'algorithmic_bytes': 226.5e6 if ops.CONV_MODE == 'h2' else None,
                'clock_ghz_under_load': H2_CONV_LAST_CLOCK_GHZ if ops.CONV_MODE == 'h2' else None,
                'mfma_pipe_utilisation_at_measured_clock': round(terms * achieved / (PEAK_BF16_MFMA_TFLOPS * H2_CONV_LAST_CLOCK_GHZ / 2.4), 4)
                if ops.CONV_MODE == 'h2' else None,
                'kernel': 'igemm_dma_kernel<%s,256,256> data gradient (%s per fp32-accurate MAC block, %s), '
                          'decoder.conv_last.0 3x3 4096->512 @64x64 N=2 (309.24 GFLOP/launch algorithmic, %.3f ms/launch, '
                          'HIP events; traffic = FETCH_SIZE*2+WRITE_SIZE from profiles/, bytes/launch)'
                          % (ops.CONV_MODE, what, inst, kt * 1e3)}
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': F32_CONV_LAST_HBM_BYTES,
            'kernel': 'igemm_conv_kernel data gradient (exact fp32 MFMA), decoder.conv_last.0 3x3 4096->512 @64x64 N=2 '
                      '(309.24 GFLOP/launch, %.3f ms/launch, HIP events)' % (kt * 1e3)}


def cpu_baseline():
    """The CPU oracle (port of the reference path over torch CPU operators) timed on this host: ONE full
    training step (fwd + loss + bwd + SGD) of the same 2x512x512 workload, all cores."""
    from oracle import semseg_oracle as O
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifests.json')))
    # torch's CPU conv/BN kernels stop scaling (and thrash) far below the 256 hardware threads of the GPU box:
    # 256 threads took 203 s for this step, so the port is timed on at most 32 threads and `cores` reports that.
    cores = min(os.cpu_count() or 1, int(os.environ.get('SEMSEG_CPU_BASELINE_THREADS', '32')))
    torch.set_num_threads(cores)
    enc = O.clone_sd(O.synth_state_dict(man['resnet50dilated'], 0), True)
    dec = O.clone_sd(O.synth_state_dict(man['ppm_deepsup@2048'], 1), True)
    img, lab = O.synth_batch(2, 512, 512, 8)
    masks = {'main': O.synth_dropout_mask(2, 512), 'deepsup': O.synth_dropout_mask(2, 512, seed=1)}
    t0 = time.perf_counter()
    res = O.segmentation_forward(enc, dec, 'resnet50dilated', 'ppm_deepsup', img, lab, training=True,
                                 dropout=masks, deep_sup_scale=0.4)
    res['loss'].backward()
    for sd in (enc, dec):
        params = {k: v for k, v in sd.items() if v.requires_grad}
        O.sgd_step(params, {k: v.grad for k, v in params.items()}, {}, 0.02)
    dt = time.perf_counter() - t0
    return {'value': round(2.0 / dt, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '1 training step (fwd+loss+bwd+SGD) of the same 2x512x512 R50dilated+PPM_deepsup batch, '
                      'torch %s CPU, no warm-up, %.1f s' % (torch.__version__, dt)}


def ddp_graph_selftest(timeout_s=150):
    """world > 1: decide whether the real run may capture its RCCL all-reduces into the step's hipGraph.  Every rank runs
    tools/probes/ddp_graph_selftest.py (a small model through exactly that code path) in a CHILD process, on its own
    rendezvous port, BEFORE this process touches the GPU or RCCL; a child that fails or does not finish in time is killed and
    counts as a failure.  Returns True if this rank's child passed (the ranks agree on the outcome afterwards)."""
    import subprocess
    import tempfile
    env = dict(os.environ)
    env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + 17)
    env.pop('TORCHELASTIC_USE_AGENT_STORE', None)       # rank 0 of the children opens its own store on the new port
    env.pop('SEMSEG_TUNE_CACHE', None)
    script = os.path.join(ROOT, 'tools', 'probes', 'ddp_graph_selftest.py')
    ok, why = False, 'not started'
    with tempfile.TemporaryFile() as log:
        try:
            p = subprocess.Popen([sys.executable, script], env=env, stdout=log, stderr=subprocess.STDOUT)
            try:
                rc = p.wait(timeout=timeout_s)
                ok, why = rc == 0, 'exit code %d' % rc
            except subprocess.TimeoutExpired:
                p.kill()                                 # the exact child started above
                p.wait()
                why = 'no result within %d s (killed)' % timeout_s
        except OSError as e:
            why = str(e)
        if not ok and env.get('RANK', '0') == '0':
            log.seek(0)
            tail = log.read()[-1500:].decode('utf-8', 'replace')
            print('[bench] data-parallel hipGraph self-test did not pass (%s); launching eagerly.\n%s' % (why, tail),
                  file=sys.stderr, flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

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
    rank, world, local = init_distributed()
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    if os.environ.get('SEMSEG_BENCH_DEVICE'):          # several ranks on one GPU (gloo): functional check of the N>1 path
        local = int(os.environ['SEMSEG_BENCH_DEVICE'])
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if selftest_ok is not None:
        flag = torch.tensor([1 if selftest_ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # every rank's child must have passed
        if int(flag.item()) == 1:
            os.environ['SEMSEG_DDP_GRAPH'] = '1'
    sm = build_model(dev)
    if world > 1:
        NativeDataParallel(sm)          # enables SyncBN statistics all-reduce over RCCL
    feed = synth_feed(dev, rank)
    step = TrainStep(sm, lr_encoder=0.02, lr_decoder=0.02, max_iters=5000 * 20,
                     graph=not args.no_graph)      # world > 1: eager unless SEMSEG_DDP_GRAPH=1 (TrainStep)

    for _ in range(args.warmup):
        loss, acc = step.step(feed)
    step.flush()                     # multi-step graphs: no staged step crosses into the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, acc = step.step(feed)
    step.flush()                     # ... and every one of the K timed steps has run before the clock stops
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

    if rank == 0:
        kt = time_dominant_kernel(dev)
        per_gpu = value / world
        out = {
            'metric': 'train images/sec (whole job) @512x512 bs2/GPU', 'value': round(value, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[ops_mode()], 'data': 'synthetic',
            'config': {'workload': 'ade20k-resnet50dilated-ppm_deepsup (BASELINE configs[1]): full train step '
                                   '(fwd+NLL loss+bwd+2xSGD), bs 2/GPU 512x512x3, 150 classes, labels 64x64',
                       'global_batch': 2 * world, 'parallelism': 'dp%d' % world,
                       'launch': 'hipGraph replay' if (step._graph is not None) else 'eager',
                       'ddp_graph_selftest': selftest_ok,
                       'conv_path': ops_mode(),
                       'images_per_sec_per_gpu': round(per_gpu, 3),
                       'step_conv_tflops_per_gpu': round(per_gpu * TRAIN_GFLOP_PER_IMG['resnet50dilated+ppm_deepsup'] * 1e-3, 2),
                       'final_loss': round(lossv, 5)},
            'roofline': roofline_entry(kt),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
