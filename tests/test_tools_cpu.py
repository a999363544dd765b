"""The profiling helpers that turn rocprofv3 counter files into the tables under profiles/ (CPU only, synthetic input)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(root, sub, counter, names, value_of):
    d = os.path.join(root, sub, 'box')
    os.makedirs(d)
    with open(os.path.join(d, sub + '_counter_collection.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        for i, name in enumerate(names, 1):
            for xcd in range(8):                       # one row per XCD, as rocprofv3 writes them
                w.writerow([i, name, counter, value_of(name) / 8.0])


def test_pmc_step_summary_finds_the_last_step_and_doubles_fetch(tmp_path):
    """three identical eager steps after some initialisation, a few read-back dispatches after the last one: the summary counts ONE
    step, doubles FETCH_SIZE (gfx950 correction of the guide) and leaves WRITE_SIZE as it is"""
    step = ['void igemm_dma_kernel<SchH2, 64, 64, 2, 2, 12>(SParams)'] * 30 + ['bn_apply_h2_kernel<true, true>(float const*, int)'] * 25 \
        + ['sgd_kernel(SgdBatch, float const*, float, float)'] * 7 + ['wprep_split_kernel(WPrepBatch, int)'] * 2
    names = ['__amd_rocclr_copyBuffer'] * 11 + ['wprep_split_kernel(WPrepBatch, int)'] * 2 + step * 3 + ['__amd_rocclr_copyBuffer'] * 2
    kib = {'void igemm': 1000.0, 'bn_apply_h': 500.0, 'sgd_kernel': 100.0, 'wprep_spli': 10.0, '__amd_rocc': 1.0}
    _write_pass(str(tmp_path), 'fetch', 'FETCH_SIZE', names, lambda n: kib[n[:10]])
    _write_pass(str(tmp_path), 'write', 'WRITE_SIZE', names, lambda n: kib[n[:10]] / 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_step_summary.py'), str(tmp_path), '10.0'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert 'fetch pass: %d dispatches, %d per step' % (len(names), len(step)) in out
    rows = {l.split()[0]: l.split() for l in out.splitlines() if l and l.split()[0] in ('igemm_dma_kernel<SchH2,', 'TOTAL')}
    total = rows['TOTAL']
    read_mb = (30 * 1000 + 25 * 500 + 7 * 100 + 2 * 10) * 2 * 1024 / 1e6
    write_mb = (30 * 1000 + 25 * 500 + 7 * 100 + 2 * 10) / 2 * 1024 / 1e6
    assert int(total[1]) == len(step)
    assert abs(float(total[2]) - read_mb) < 0.06 and abs(float(total[3]) - write_mb) < 0.06
    assert 'TB/s average over the step' in out


def test_abi_call_trace_lists_every_convolution_of_the_step():
    """tools/abi_call_trace.py: the C-ABI call sequence of one configs[1] training step on the recording stub -- 63 forward convs
    (five of them in the Winograd domain), as many data gradients minus the stem's, every weight gradient; the geometry columns are what the PMC tables are labelled with"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'abi_call_trace.py'), '--config', '1', '--grep', 'conv2d_fwd,conv2d_dgrad,conv2d_wgrad,winograd_gemm,winograd_wgrad_gemm'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in r.stdout.splitlines() if l.strip()]
    count = lambda name: sum(1 for l in lines if l[0] == name)        # noqa: E731
    fwd = count('conv2d_fwd_stats_h2') + count('conv2d_fwd_h2')
    wino_gemm, wino_wgrad = count('winograd_gemm_h2'), count('winograd_wgrad_gemm_h2')
    # the data gradient of a Winograd layer is either the batched GEMM (+ output transform) or the fused kernel -- which one is a measured
    # choice the trace takes from the shipped performance database (conv_last: fused)
    wino_fused = count('winograd_gemm_output_h2')
    wino_layers = (wino_gemm + wino_fused) // 2
    convs = (3 + 16 * 3 + 4) + (4 + 1 + 1) + (1 + 1)     # deep-stem ResNet-50 (stem, 16 bottlenecks, 4 downsamples), PPM + conv_last + classifier, deepsup
    assert (wino_gemm + wino_fused) % 2 == 0 and wino_layers == 5, (wino_gemm, wino_fused)
    assert fwd + wino_layers == convs, (fwd, wino_gemm, wino_fused)    # a Winograd layer runs a GEMM in the forward and one in the data gradient
    assert count('conv2d_dgrad_h2') + wino_layers == convs - 1        # no data gradient into the image
    # inside TrainStep: slabs + one multi-tensor reduce; the small weight gradients (plan on the 64 x 64 tile) wait for one batched call
    batched = sum(int(l[1]) for l in lines if l[0] == 'conv2d_wgrad_multi_h2')
    assert count('conv2d_wgrad_multi_h2') <= 1
    assert count('conv2d_wgrad_h2') + count('conv2d_wgrad_slabs_h2') + batched + wino_wgrad == convs
    stem = [l for l in lines if l[0] == 'conv2d_fwd_stats_h2'][0]
    assert stem[2:12] == ['2', '512', '512', '3', '64', '3', '3', '2', '1', '1']           # after y_ld: N H W C K R S stride pad dil


def test_shape_stream_simulator_two_clocks():
    """tools/shape_stream_sim.simulate on a hand-made stream: the host issues, the device executes; a recording pass behind a device
    synchronize is exposed, one without it hides behind the replays already queued; the LRU evicts and re-records"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import shape_stream_sim as S
    A, B, C = (2, 64, 64), (2, 72, 104), (2, 96, 64)
    costs = dict(replay_ms=10.0, eager_ms=20.0, capture_ms=30.0, host_replay_ms=0.0)
    # A is known; B appears in the window after three replays of A: 3 x 10 ms of device work are queued when its recording starts
    stream = [A, A, A, A, B, B]
    v_sync, ev = S.simulate(stream, 1, 5, cap=8, first_sight=True, sync_capture=True, **costs)
    v_free, ev2 = S.simulate(stream, 1, 5, cap=8, first_sight=True, sync_capture=False, **costs)
    assert ev == ev2 == {'replayed': 4, 'captured': 1}
    assert abs(2e3 * 5 / v_sync - (3 * 10 + 30 + 2 * 10)) < 1e-6        # recording starts when the device has drained, then two steps
    assert abs(2e3 * 5 / v_free - (5 * 10)) < 1e-6                      # 30 ms of recording behind 30 ms of queued replays: hidden
    # second sight: the warm-up step ran A eagerly, so A is recorded inside the window too; B costs an eager pass and a recording
    v_old, ev3 = S.simulate(stream, 1, 5, cap=8, first_sight=False, sync_capture=True, **costs)
    assert ev3 == {'replayed': 2, 'eager': 1, 'captured': 2} and v_old < v_sync
    # an LRU of one graph re-records on every change of shape
    _, ev4 = S.simulate([A, B, A, B, A, B], 0, 6, cap=1, first_sight=True, sync_capture=False, **costs)
    assert ev4['captured'] == 6 and ev4['evicted'] == 5
