"""Host-side launch-plan tuner for the operand-split (h2 / s3) convolution kernels.

"Measure, don't guess": the best (tile, split-K) of an implicit-GEMM conv on MI355X depends on how the grid
quantises over 256 CUs x resident-block slots and on L2 behaviour of the particular geometry; the built-in
heuristic of the library (csrc/conv_split.hip plan_gemm/plan_wgrad) is within ~25 % of the best plan on average and
off by 2x on some layers (profiles/r1c_conv_bench_s3_sweep.txt).  The first time a conv geometry is seen
OUTSIDE hipGraph capture, every candidate plan is timed with HIP events on the real buffers and the winner is
pinned in the library with semseg_conv2d_{h2,s3}_set_plan.  Plans only change the fp32 summation order (split-K),
never the arithmetic.  Disable with SEMSEG_TUNE=0.
"""
import os

import torch

from . import _native

ENABLED = os.environ.get('SEMSEG_TUNE', '1') != '0'
# optional plan cache (JSON): plans tuned by one process are pinned without re-timing by the next (e.g. a profiled run
# after a tuned run on the same box).  Unset -> no file is read or written.
CACHE = os.environ.get('SEMSEG_TUNE_CACHE', '')
# Shipped performance database (round 4): launch plans MEASURED by this tuner on an MI355X with the library of this tree, for the
# conv geometries of the BASELINE configurations (mit_semseg/perfdb/gfx950_h2.json; regenerate with tools/make_perfdb.sh).  It is
# read before anything is timed, so a fresh process starts from measured plans instead of re-timing ~200 candidates per geometry
# (the driver's bench run and every test process did), and two runs of one tree launch the same plans -- the run-to-run spread of
# the tuner's own choices no longer enters the number.  A geometry that is not in it is timed as before.  SEMSEG_TUNE_DB=0 ignores
# the file (everything is timed afresh); SEMSEG_TUNE_CACHE still names a read-write cache that takes precedence.
PERFDB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'perfdb', 'gfx950_h2.json')
USE_PERFDB = os.environ.get('SEMSEG_TUNE_DB', '1') != '0'
_done = {}          # (scheme, pass, geom) -> (tile, split, ms)
_cache_loaded = False
stats_db = {'from_perfdb': 0, 'from_cache': 0}
# Variable-size batches (BASELINE configs[3]: every per-GPU batch has its own H x W) would put a full tile x split sweep of every
# conv of the network in front of every new shape.  The winning plan of a layer depends on its channel configuration and on how
# many output pixels it has, not on how they are arranged, so a geometry that has not been seen inherits the plan tuned for the
# same layer in the same HALF-OCTAVE of pixel count (round(2 log2 M)); only the first geometry of a bucket is timed.
# SEMSEG_TUNE_BUCKETS=0: time every geometry.
BUCKETS = os.environ.get('SEMSEG_TUNE_BUCKETS', '1') != '0'
_bucket_plans = {}  # (scheme, pass, C, K, R, S, stride, pad, dil, bucket) -> (tile, split)
stats = {'timed': 0, 'inherited': 0, 'missed_capturing': 0}   # missed_capturing: geometries met inside a graph capture with no plan to inherit
# launches are being RECORDED, not issued (ops.BranchesFn: side-by-side launches of independent sub-networks, csrc/batch.h): nothing
# can be timed -- treated like a graph capture (a plan may be inherited, a geometry without one runs on the library's default)
NO_TIMING = [False]


def _cannot_time():
    return NO_TIMING[0] or torch.cuda.is_current_stream_capturing()


def _bucket_key(scheme, pass_id, geom):
    import math
    n, h, w, c, k, r, s, stride, pad, dil = geom
    oh = (h + 2 * pad - dil * (r - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (s - 1) - 1) // stride + 1
    m = max(1, n * oh * ow)
    return (scheme, pass_id, c, k, r, s, stride, pad, dil, int(round(2.0 * math.log2(m))))


def _bucket_lookup(bkey):
    """the plan of bucket `bkey`, else of the NEAREST half-octave of the same layer (up to two octaves away): a plan measured for the
    same layer at a neighbouring size beats timing ~200 candidates inside a training step -- the variable-size stream of BASELINE
    configs[3] keeps meeting new pixel counts (81 geometries were timed inside 400 steps of the raw stream before this, gpurun r6N)"""
    rec = _bucket_plans.get(bkey)
    if rec is not None:
        return rec
    for d in (1, -1, 2, -2, 3, -3, 4, -4):
        rec = _bucket_plans.get(bkey[:-1] + (bkey[-1] + d,))
        if rec is not None:
            return rec
    return None


def _bucket_key_wino(pass_id, geom):
    """pass 3 (tile of the batched Winograd GEMM, geom = (tiles, 1, 1, C, K, ...)) and pass 4 (launch form, geom = (tiles, n, dil, C, K,
    ...)): the same layer in the same half-octave of Winograd tiles.  Without it every new batch shape of the variable-size stream
    (BASELINE configs[3]) timed ~10 GEMM tiles and ~10 fused forms of conv_last / cbr_deepsup / layer4 on its first sight: 0.3 - 0.5 s
    per shape (bench.py raw_stream leg, gpurun r6M)."""
    import math
    tiles = max(1, int(geom[0]))
    if pass_id == 3:
        return ('w3', int(geom[3]), int(geom[4]), int(round(2.0 * math.log2(tiles))))
    return ('w4', int(geom[2]), int(geom[3]), int(geom[4]), int(round(2.0 * math.log2(tiles))))


def _load_cache():
    """perf database first (read-only), then the read-write cache of SEMSEG_TUNE_CACHE on top of it"""
    global _cache_loaded
    _cache_loaded = True
    import json
    L = _native.lib()
    for path, counter in (((PERFDB if USE_PERFDB else ''), 'from_perfdb'), (CACHE, 'from_cache')):
        if not path or not os.path.exists(path):
            continue
        try:
            entries = json.load(open(path))
        except ValueError:
            continue
        for k, v in entries.items():
            if k.startswith('_'):
                continue
            parts = k.split(',')
            key = (parts[0],) + tuple(int(t) for t in parts[1:])
            if key[1] == 4:                   # a choice between launch forms (choose()), not a library plan
                if counter == 'from_perfdb' and os.environ.get('SEMSEG_TUNE_RECHOOSE', '0') == '1':
                    continue                  # tools/rechoose_forms.sh: new launch forms exist, every choice is timed afresh
                _done[key] = tuple(v)
                stats_db[counter] += 1
                bk = _bucket_key_wino(4, key[2:])
                if counter == 'from_cache' or bk not in _bucket_plans:
                    _bucket_plans[bk] = (int(v[0]), int(v[1]))
                continue
            if v[0] >= 0 and _set_plan(L, key[0])(key[1], *key[2:], int(v[0]), int(v[1])) != 0:
                continue                      # a plan this build of the library does not know: the geometry is timed afresh
            _done[key] = tuple(v)
            stats_db[counter] += 1
            if key[1] == 3:
                bk = _bucket_key_wino(3, key[2:])
                if counter == 'from_cache' or bk not in _bucket_plans:
                    _bucket_plans[bk] = (int(v[0]), int(v[1]))
            if key[1] in (0, 1, 2):
                bk = _bucket_key(key[0], key[1], key[2:])
                if counter == 'from_cache' or bk not in _bucket_plans:      # the later source (the user's cache) wins, as for _done;
                    _bucket_plans[bk] = (int(v[0]), int(v[1]))               # within one source the first entry of a bucket stays


def _save_cache():
    if not CACHE:
        return
    import json
    tmp = CACHE + '.tmp.%d' % os.getpid()
    with open(tmp, 'w') as f:
        json.dump({','.join(map(str, k)): list(v) for k, v in _done.items()}, f)
    os.replace(tmp, CACHE)
_SPLITS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64)
_WSPLITS = _SPLITS
# fwd/dgrad tile ids per scheme (csrc/conv_split.hip): 0..2 register staged, 3 = 256x128 LDS-DMA, h2 only: 4 = 256x128
# 3-slot ring, 5 = 256x256, 6..10 their software-pipelined / 128x128 forms, 11..13 = 4-wave forms (256x256, 256x128 2- and 3-slot),
# 14 = 256x256 on 16 waves, 15..17 = 64x64 / 128x64 on the LDS-DMA ring (4 waves).  SEMSEG_TUNE_TILES=0,1,2,3 restricts the candidates (e.g. to bisect a suspect kernel).
_TILES = {'s3': (0, 1, 2, 3), 'h2': tuple(range(27))}      # 19..21: 4- / 5-slot rings on the small tiles; 22..24: 64-deep k-tiles; 25 / 26: 256 x 256 on the ring of five half tiles
# weight-gradient tile ids: 0 = 128x128, 1 = 64x64 register staged; h2 only: 2 = 128x128 LDS-DMA, 3 = 256x128 LDS-DMA ring
_WTILES = {'s3': (0, 1), 'h2': tuple(range(15))}      # 10: all nine taps of a 3x3 stride-1 conv in one block; 11 ... 14: ring of five half tiles
_ALLOW = os.environ.get('SEMSEG_TUNE_TILES', '')
if _ALLOW:
    _allow = tuple(int(t) for t in _ALLOW.split(','))
    _TILES = {k: tuple(t for t in v if t in _allow) for k, v in _TILES.items()}
_WALLOW = os.environ.get('SEMSEG_TUNE_WTILES', '')          # the same for the weight-gradient tiles
if _WALLOW:
    _wallow = tuple(int(t) for t in _WALLOW.split(','))
    _WTILES = {k: tuple(t for t in v if t in _wallow) for k, v in _WTILES.items()}


def _set_plan(L, scheme):
    return getattr(L, 'semseg_conv2d_%s_set_plan' % scheme)


def _time(launch, reps):
    """ms per launch: the best of two rounds of `reps` back-to-back launches; short kernels get more launches per round
    (a 2-launch measurement of a 20 us kernel is +-10 %, enough to pin the wrong plan)"""
    best = float('inf')
    for _ in range(2):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(reps):
            launch()
        en.record()
        en.synchronize()
        ms = st.elapsed_time(en) / reps
        best = min(best, ms)
        reps = max(reps, min(24, int(0.4 / max(ms, 1e-3))))      # second round: ~0.4 ms of work
    return best


def tuned_plans():
    return dict(_done)


_TIMING = [False]


def timing():
    """True while ensure() is timing candidates: a launch closure may then add the follow-up work that only SOME plans need
    (ops.ConvBNActFn: the statistics sweep of a split-K plan) so that the candidates are compared on what they really cost"""
    return _TIMING[0]


WINO_TILES = (6, 7, 8, 9, 10, 14, 22, 24, 25, 26)       # tile forms the batched Winograd forward GEMM can run on (csrc/conv_split.hip, plan pass 3)


def ensure_winograd_gemm(tiles, c, k, launch):
    """launch plan of the batched GEMM of the Winograd forward (semseg_winograd_gemm_h2): the tile form is pinned per
    (tiles, C, K) as plan pass 3, timed once on the real buffers like the direct convolutions (the library's own rule -- the
    8-wave 256 x 256 tile -- is 15-20 % behind the 16-wave form on the 512- and 1024-channel layers,
    profiles/r4_winograd_midsize_probe.txt)"""
    geom = (int(tiles), 1, 1, int(c), int(k), 3, 3, 1, 1, 1)
    key = ('h2', 3) + geom
    if ENABLED and not _cache_loaded:
        _load_cache()
    if not ENABLED or key in _done:
        return
    L = _native.lib()
    set_plan = _set_plan(L, 'h2')
    bkey = _bucket_key_wino(3, geom)
    if BUCKETS and _bucket_lookup(bkey) is not None:
        tile, _ = _bucket_lookup(bkey)
        if tile < 0 or set_plan(3, *geom, tile, 1) == 0:
            _done[key] = (tile, 1 if tile >= 0 else 0, None)
            stats['inherited'] += 1
            return
    if not torch.cuda.is_available():
        return
    if _cannot_time():
        stats['missed_capturing'] += 1          # runs on the library's default; engine.TrainStep re-captures after an eager (timing) pass
        return
    stats['timed'] += 1
    best = None
    _TIMING[0] = True
    try:
        launch()
        best = (-1, 0, _time(launch, 2))
        for tile in WINO_TILES:
            if set_plan(3, *geom, tile, 1) != 0:
                continue
            try:
                launch()
                ms = _time(launch, 3)
            except RuntimeError:
                continue
            if ms < best[2]:
                best = (tile, 1, ms)
    finally:
        _TIMING[0] = False
        if best is None or best[0] < 0:
            set_plan(3, *geom, -1, 0)
        else:
            set_plan(3, *geom, best[0], 1)
    _done[key] = best
    if best is not None:
        _bucket_plans.setdefault(bkey, (int(best[0]), int(best[1])))
    _save_cache()


def choose(geom, candidates, default=0):
    """which of several launch FORMS of one pass runs (e.g. the Winograd data gradient as batched GEMM + output transform, or as
    the fused kernel on a 3- / 4-slot ring): every candidate -- a closure that issues the whole alternative on the current stream --
    is timed once per geometry on the real buffers, outside graph capture, and the index of the fastest is remembered as plan
    pass 4 (host-side only; the library has no plan for it).  Returns the index to run."""
    key = ('h2', 4) + tuple(int(g) for g in geom)
    if ENABLED and not _cache_loaded:
        _load_cache()
    rec = _done.get(key)
    if rec is not None:
        return int(rec[0]) if 0 <= int(rec[0]) < len(candidates) else default
    bkey = _bucket_key_wino(4, key[2:])
    near = _bucket_lookup(bkey) if (ENABLED and BUCKETS) else None
    if near is not None and 0 <= near[0] < len(candidates):
        _done[key] = (near[0], 1, None)          # the form measured for the same layer at a neighbouring size
        stats['inherited'] += 1
        return near[0]
    if not ENABLED or not torch.cuda.is_available():
        return default
    if _cannot_time():
        stats['missed_capturing'] += 1
        return default
    stats['timed'] += 1
    best = None
    for i, fn in enumerate(candidates):
        try:
            fn()
            ms = min(_time(fn, 6), _time(fn, 6))        # forms a few per cent apart: four rounds each, best one counts
        except RuntimeError:
            continue
        if best is None or ms < best[2]:
            best = (i, 1, ms)
    if best is None:
        return default
    _done[key] = best
    _bucket_plans.setdefault(bkey, (int(best[0]), 1))
    _save_cache()
    return best[0]


SEEN = None          # tools/step_plan_refine.py: a set that collects the plan keys the running model asks for
RANKED = None        # ... and a dict key -> the fastest candidates of a timing sweep [(ms, tile, split), ...]


def ensure(scheme, pass_id, geom, launch):
    """scheme = 'h2' | 's3'; geom = (N,H,W,C,K,R,S,stride,pad,dil); launch() issues the conv of this pass on the
    current stream."""
    key = (scheme, pass_id) + tuple(geom)
    if SEEN is not None:
        SEEN.add(key)
    if ENABLED and not _cache_loaded:
        _load_cache()
    if not ENABLED or key in _done:
        return
    if not torch.cuda.is_available():
        return                                  # host-logic dry runs (CPU, stubbed ABI): never time
    L = _native.lib()
    n, h, w, c, k, r, s, stride, pad, dil = geom
    oh = (h + 2 * pad - dil * (r - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (s - 1) - 1) // stride + 1
    # reduction length in k-tiles (fwd/dgrad: taps x 32-channel chunks; wgrad: 32-pixel chunks)
    if pass_id == 0:
        kt = r * s * ((c + 31) // 32)
    elif pass_id == 1:
        kt = r * s * ((k + 31) // 32)
    else:
        kt = (n * oh * ow + 31) // 32
    tiles = _WTILES[scheme] if pass_id == 2 else _TILES[scheme]
    set_plan = _set_plan(L, scheme)
    bkey = _bucket_key(scheme, pass_id, geom)
    if BUCKETS and _bucket_lookup(bkey) is not None:
        tile, split = _bucket_lookup(bkey)
        ok = True
        if tile >= 0:
            while split > 1 and kt // split < 4:              # the same validity rule as the sweep below
                split = max(s for s in _SPLITS if s < split)
            ok = set_plan(pass_id, *geom, tile, split) == 0   # refused: this geometry is not one the inherited kernel takes
        if ok:
            _done[key] = (tile, split if tile >= 0 else 0, None)
            stats['inherited'] += 1
            return
    if _cannot_time():
        stats['missed_capturing'] += 1
        return                                  # graph capture: a plan may be inherited (above), never timed
    stats['timed'] += 1
    best = None
    ranked = []                                     # (ms, tile, split) of every candidate that ran
    _TIMING[0] = True
    try:
        launch()                                    # heuristic plan first (also warms caches / sizes the workspace)
        base = _time(launch, 2)
        best = (-1, 0, base)
        ranked.append((base, -1, 0))
        for tile in tiles:
            splits = _SPLITS
            if pass_id == 2:
                # the all-taps tile (10) has 9x fewer tiles than the per-tap kernels: its split over the pixels fills the chip
                splits = (16, 32, 64, 96, 128, 192, 256, 384, 512) if tile == 10 else _WSPLITS
            for split in splits:
                if split > 1 and kt // split < 4:
                    break
                if set_plan(pass_id, *geom, tile, split) != 0:
                    break                                   # the library refuses this tile for this geometry
                try:
                    launch()
                    ms = _time(launch, 2)
                except RuntimeError:
                    continue
                ranked.append((ms, tile, split))
                if ms < best[2]:
                    best = (tile, split, ms)
                if ms > 3.0 * best[2] and split >= 4 and not (pass_id == 2 and tile == 10):
                    break
        # play-off: with ~200 candidates per geometry a 5 % timing outlier picks the wrong plan now and then; the three
        # fastest are timed again (more launches per round) and the best mean of the two measurements wins
        ranked.sort()
        if RANKED is not None:
            RANKED[key] = ranked[:6]
        finals = []
        for ms, tile, split in ranked[:3]:
            if tile < 0:
                set_plan(pass_id, *geom, -1, 0)
            else:
                _native.check(set_plan(pass_id, *geom, tile, split), 'set_plan')
            try:
                launch()
                finals.append((0.5 * (ms + _time(launch, 4)), tile, split))
            except RuntimeError:
                continue
        if finals:
            ms, tile, split = min(finals)
            best = (tile, split, ms)
    finally:
        _TIMING[0] = False
        if best is None or best[0] < 0:
            set_plan(pass_id, *geom, -1, 0)
        else:
            set_plan(pass_id, *geom, best[0], best[1])
    _done[key] = best
    if best is not None:
        _bucket_plans[bkey] = (best[0], best[1])
    _save_cache()
