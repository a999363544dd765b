"""Parity of every HIP operator against the torch CPU fp32/fp64 operator the reference calls
(SURVEY 2b table).  Runs on the MI355X only (-m gpu); all calls go through the C ABI via mit_semseg.ops."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# fp32 MFMA == fmaf chain: error vs an fp64 reference is fp32-roundoff class (guide: 1e-7 * sum|a*b|)
REL = 2e-5


def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda:0')


def rel_err(got, ref):
    ref = ref.double()
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def cl(x):
    """NCHW cpu tensor -> cuda tensor in NHWC memory"""
    return x.to(dev()).contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # (N, C, H, W, K, ksize, stride, pad, dil, bias)
    (2, 64, 32, 32, 64, 3, 1, 1, 1, False),      # plain 3x3
    (2, 256, 16, 16, 256, 3, 1, 2, 2, False),    # dilated d2 (layer3)
    (1, 512, 16, 16, 512, 3, 1, 4, 4, False),    # dilated d4 (layer4)
    (2, 3, 64, 64, 64, 3, 2, 1, 1, False),       # stem: C=3, stride 2 (scalar-load path)
    (2, 128, 33, 31, 256, 3, 2, 1, 1, False),    # stride 2, odd sizes
    (2, 1024, 16, 16, 256, 1, 1, 0, 1, False),   # 1x1
    (2, 256, 17, 17, 512, 1, 2, 0, 1, False),    # 1x1 stride 2 (downsample)
    (2, 512, 16, 16, 150, 1, 1, 0, 1, True),     # classifier: K=150 + bias
    (2, 2048, 1, 1, 512, 1, 1, 0, 1, False),     # PPM scale 1: M=2
    (2, 2048, 6, 6, 512, 1, 1, 0, 1, False),     # PPM scale 6 (split-K)
    (2, 48, 24, 24, 48, 3, 1, 1, 1, False),      # HRNet 48 channels (BK=16 path)
    (1, 180, 12, 12, 150, 1, 1, 0, 1, True),     # C1 classifier: C=180 (partial chunk)
    (1, 720, 12, 12, 180, 3, 1, 1, 1, False),    # C1 cbr
    (2, 4096, 8, 8, 512, 3, 1, 1, 1, False),     # conv_last shape (long K, split-K)
    (2, 96, 20, 20, 192, 3, 2, 1, 1, False),     # HRNet exchange stride 2
]


@pytest.mark.parametrize('mode', ['h2', 's3', 'f32'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_fwd_bwd(case, mode, monkeypatch):
    """all conv families through the autograd op: split-fp16 MFMA (h2, default), split-bf16 MFMA (s3) and exact-fp32
    MFMA (f32), same fp32-class tolerance against a float64 CPU convolution"""
    from mit_semseg import ops
    monkeypatch.setattr(ops, 'CONV_MODE', mode)
    n, c, h, w, k, ks, stride, pad, dil, bias = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, ks, ks, generator=g) / (c * ks * ks) ** 0.5
    b = torch.randn(k, generator=g) if bias else None
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())

    xg, wg = cl(x).requires_grad_(True), cl(wt).requires_grad_(True)
    bg = b.to(dev()).requires_grad_(True) if bias else None
    y = ops.conv2d(xg, wg, bg, stride, pad, dil)
    assert y.shape == yr.shape
    y.backward(cl(gy))
    torch.cuda.synchronize()
    assert rel_err(y, yr) < REL, ('fwd', rel_err(y, yr))
    assert rel_err(xg.grad, xr.grad) < REL, ('dgrad', rel_err(xg.grad, xr.grad))
    assert rel_err(wg.grad, wr.grad) < REL * 4, ('wgrad', rel_err(wg.grad, wr.grad))
    if bias:
        assert rel_err(bg.grad, br.grad) < REL, ('bgrad', rel_err(bg.grad, br.grad))


@pytest.mark.parametrize('mode', ['h2', 's3', 'f32'])
def test_conv2d_reads_channel_slice(mode, monkeypatch):
    """x given as a channel slice of a wider NHWC buffer (ld > C), as the concat consumers do"""
    from mit_semseg import ops
    monkeypatch.setattr(ops, 'CONV_MODE', mode)
    g = torch.Generator().manual_seed(1)
    big = torch.randn(2, 96, 10, 10, generator=g)
    wt = torch.randn(32, 64, 3, 3, generator=g) / 24
    yr = F.conv2d(big[:, 16:80].double(), wt.double(), None, 1, 1, 1)
    y = ops.conv2d(cl(big)[:, 16:80], cl(wt), None, 1, 1, 1)
    assert rel_err(y, yr) < REL


BN_CASES = [(2, 64, 16, 16), (2, 48, 9, 7), (1, 512, 1, 2), (2, 2048, 8, 8), (2, 180, 5, 5)]


@pytest.mark.parametrize('shape', BN_CASES, ids=str)
@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('relu,res', [(False, False), (True, False), (True, True)])
def test_batch_norm_act(shape, training, relu, res):
    from mit_semseg import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    r = torch.randn(shape, generator=g) if res else None
    mom = 0.1
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, training, mom, 1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    gy = torch.randn(shape, generator=g)
    yr.backward(gy.double())

    xg = cl(x).requires_grad_(True)
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    rg = cl(r).requires_grad_(True) if res else None
    rmg, rvg = rm.to(dev()), rv.to(dev())
    y = ops.batch_norm_act(xg, gg, bg, rmg, rvg, residual=rg, training=training, momentum=mom, eps=1e-5, relu=relu)
    y.backward(cl(gy))
    torch.cuda.synchronize()
    tol = 2e-5
    assert rel_err(y, yr) < tol
    # 2 values per channel: xhat = +-1 and dx is pure cancellation (|dx| ~ eps/var * |dy|): compare loosely there
    assert rel_err(xg.grad, xr.grad) < (1e-4 if n * h * w > 2 else 2e-3)
    assert rel_err(gg.grad, gr.grad) < 1e-4 and rel_err(bg.grad, br.grad) < 1e-4
    if res:
        assert rel_err(rg.grad, rr.grad) < tol
    assert rel_err(rmg, rmr) < 1e-5 and rel_err(rvg, rvr) < 1e-5


@pytest.mark.parametrize('kind,shape,affine,training', [
    ('1d', (16, 10), True, True), ('1d', (16, 10), True, False),          # testSyncBatchNormNormalTrain / NormalEval
    ('1d', (16, 10), False, True), ('1d', (16, 10), False, False),        # testSyncBatchNormSyncTrain / SyncEval: affine=False
    ('2d', (16, 10, 16, 16), True, True),                                 # testSyncBatchNorm2DSyncTrain
    ('2d', (4, 12, 8, 8), False, True),
])
def test_sync_batchnorm_modules_like_the_reference_unit_tests(kind, shape, affine, training):
    """the cases of the reference's own SyncBN tests (lib/nn/modules/tests/test_sync_batchnorm.py:44-107): output, input
    gradient and running statistics against nn.BatchNorm with the same arguments, 10 features (not a multiple of the kernels'
    4-channel lanes), affine=False included (batchnorm.py:39-48,79-83)"""
    import torch.nn as nn
    from mit_semseg.lib.nn import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d
    c = shape[1]
    ref = (nn.BatchNorm1d if kind == '1d' else nn.BatchNorm2d)(c, eps=1e-5, momentum=0.001, affine=affine).double()
    mine = (SynchronizedBatchNorm1d if kind == '1d' else SynchronizedBatchNorm2d)(c, eps=1e-5, momentum=0.001, affine=affine)
    assert ('weight' in mine.state_dict()) == affine and ('bias' in mine.state_dict()) == affine
    mine.to(dev())
    ref.train(training)
    mine.train(training)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(shape, generator=g)
    xr = x.double().requires_grad_(True)
    xg = x.to(dev()).requires_grad_(True)
    yr = ref(xr)
    yr.sum().backward()
    y = mine(xg)
    y.sum().backward()
    torch.cuda.synchronize()
    assert y.shape == yr.shape
    assert (y.cpu().double() - yr).abs().max().item() < 1e-5
    assert (xg.grad.cpu().double() - xr.grad).abs().max().item() < 1e-5
    assert (mine.running_mean.cpu().double() - ref.running_mean).abs().max().item() < 1e-6
    assert (mine.running_var.cpu().double() - ref.running_var).abs().max().item() < 1e-6


def test_batch_norm_train_single_value_raises():
    from mit_semseg import ops
    x = torch.randn(1, 8, 1, 1, device=dev())
    o = torch.ones(8, device=dev())
    with pytest.raises(ValueError, match='more than 1 value per channel'):
        ops.batch_norm_act(x, o, o, o.clone(), o.clone(), training=True)


@pytest.mark.parametrize('shape', [(2, 128, 32, 32), (1, 64, 17, 13), (2, 4, 5, 5)], ids=str)
def test_maxpool(shape):
    from mit_semseg import ops
    g = torch.Generator().manual_seed(3)
    x = F.relu(torch.randn(shape, generator=g))         # ties at 0 as after ReLU in the stem
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = cl(x).requires_grad_(True)
    y = ops.max_pool_3x3_s2(xg)
    y.backward(cl(gy))
    assert torch.equal(y.cpu(), yr)
    torch.testing.assert_close(xg.grad.cpu().contiguous(), xr.grad, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize('h,w,s', [(64, 64, 1), (64, 64, 2), (64, 64, 3), (64, 64, 6), (16, 16, 6), (9, 13, 3), (2, 2, 6),
                                   (4, 4, 3)])
def test_adaptive_avg_pool(h, w, s):
    from mit_semseg import ops
    g = torch.Generator().manual_seed(s)
    x = torch.randn(2, 72, h, w, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, s)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xg = cl(x).requires_grad_(True)
    y = ops.adaptive_avg_pool(xg, s)
    y.backward(cl(gy))
    assert rel_err(y, yr) < 1e-6
    assert rel_err(xg.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('ih,iw,oh,ow', [(1, 1, 64, 64), (2, 2, 64, 64), (3, 3, 64, 64), (6, 6, 64, 64), (16, 16, 32, 32),
                                         (8, 8, 35, 45), (32, 32, 128, 128), (9, 13, 4, 5), (64, 64, 70, 90), (6, 6, 16, 16)])
def test_bilinear(ih, iw, oh, ow):
    from mit_semseg import ops
    g = torch.Generator().manual_seed(ih * 7 + ow)
    x = torch.randn(2, 48, ih, iw, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, size=(oh, ow), mode='bilinear', align_corners=False)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xg = cl(x).requires_grad_(True)
    y = ops.interpolate_bilinear(xg, (oh, ow))
    y.backward(cl(gy))
    # source coordinates are computed in fp32 (as torch does for fp32 tensors); the fp64 reference differs by that
    assert rel_err(y, yr) < 1e-5
    assert rel_err(xg.grad, xr.grad) < 2e-5
    y32 = F.interpolate(x, size=(oh, ow), mode='bilinear', align_corners=False)
    assert rel_err(y, y32) < 1e-6


def test_bilinear_150_classes():
    """inference branch (models.py:480-484): logits with C=150 (not a multiple of 4) are up-sampled to segSize"""
    from mit_semseg import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 150, 8, 10, generator=g)
    yr = F.interpolate(x, size=(35, 45), mode='bilinear', align_corners=False)
    y = ops.interpolate_bilinear(cl(x), (35, 45))
    assert rel_err(y, yr) < 1e-6


def test_bilinear_accumulate_relu():
    from mit_semseg import ops
    g = torch.Generator().manual_seed(11)
    x, base = torch.randn(2, 48, 8, 8, generator=g), torch.randn(2, 48, 16, 16, generator=g)
    xr, br = x.double().requires_grad_(True), base.double().requires_grad_(True)
    yr = F.relu(br + F.interpolate(xr, size=(16, 16), mode='bilinear', align_corners=False))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xg, bg = cl(x).requires_grad_(True), cl(base).requires_grad_(True)
    y = ops.interpolate_bilinear(xg, (16, 16), base=bg, relu=True)
    y.backward(cl(gy))
    assert rel_err(y, yr) < 2e-6 and rel_err(xg.grad, xr.grad) < 1e-5 and rel_err(bg.grad, br.grad) < 1e-6


def test_concat_add_scale():
    from mit_semseg import ops
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(2, 64, 6, 6, generator=g), torch.randn(2, 32, 6, 6, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    m = torch.rand(2, 96, generator=g)
    yr = F.relu(torch.cat([ar, br], 1) * m[:, :, None, None] + 1.0)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    ag, bg = cl(a).requires_grad_(True), cl(b).requires_grad_(True)
    one = torch.ones(2, 96, 6, 6, device=dev()).contiguous(memory_format=torch.channels_last)
    y = ops.add_act(ops.scale_nc(ops.concat([ag, bg]), m.to(dev())), one, relu=True)
    y.backward(cl(gy))
    torch.testing.assert_close(y.cpu().contiguous(), yr.detach(), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(ag.grad.cpu().contiguous(), ar.grad, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(bg.grad.cpu().contiguous(), br.grad, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize('n,h,w', [(2, 8, 8), (1, 48, 48), (2, 64, 64)])
def test_log_softmax_nll_acc(n, h, w):
    from mit_semseg import ops
    g = torch.Generator().manual_seed(h)
    z = torch.randn(n, 150, h, w, generator=g) * 3
    lab = torch.randint(-1, 150, (n, h, w), generator=g)
    zr = z.double().requires_grad_(True)
    lp = F.log_softmax(zr, dim=1)
    loss_r = F.nll_loss(lp, lab, ignore_index=-1)
    (loss_r * 0.4).backward()
    pr = lp.max(dim=1)[1]
    valid = (lab >= 0)
    acc_r = (valid & (pr == lab)).sum().float() / (valid.sum().float() + 1e-10)

    zg = cl(z).requires_grad_(True)
    logp = ops.log_softmax(zg)
    loss, acc = ops.nll_loss_acc(logp, lab.to(dev()), ignore_index=-1)
    (loss * 0.4).backward()
    assert rel_err(logp, lp) < 1e-6
    assert abs(loss.item() - loss_r.item()) < 1e-5
    assert abs(acc.item() - acc_r.item()) < 1e-7
    assert rel_err(zg.grad, zr.grad) < 1e-5
    prob = ops.softmax(cl(z))
    assert rel_err(prob, F.softmax(z.double(), dim=1)) < 1e-6


def test_nll_all_ignored_is_nan():
    from mit_semseg import ops
    z = torch.randn(1, 150, 4, 4, device=dev()).contiguous(memory_format=torch.channels_last)
    lab = torch.full((1, 4, 4), -1, dtype=torch.long, device=dev())
    loss, acc = ops.nll_loss_acc(ops.log_softmax(z), lab)
    assert torch.isnan(loss).item() and acc.item() == 0.0


def test_sgd_step_matches_torch():
    from mit_semseg import ops
    g = torch.Generator().manual_seed(9)
    ps = [torch.randn(64, 32, 3, 3, generator=g), torch.randn(77, generator=g), torch.randn(5, 3, 1, 1, generator=g)]
    wds = [1e-4, 0.0, 1e-4]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opts = [torch.optim.SGD([r], lr=0.02, momentum=0.9, weight_decay=wd) for r, wd in zip(ref, wds)]
    gp = [p.to(dev()) for p in ps]
    bufs = [torch.zeros_like(p) for p in gp]
    lr = torch.tensor([0.02], device=dev())
    for it in range(3):
        grads = [torch.randn(p.shape, generator=g) for p in ps]
        for r, gr, o in zip(ref, grads, opts):
            r.grad = gr.clone()
            o.step()
        ops.sgd_step(gp, [x.to(dev()) for x in grads], bufs, it == 0, wds, lr, 0.9, 1.0)
    for p, r in zip(gp, ref):
        torch.testing.assert_close(p.cpu(), r.detach(), atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------------------------------------
# h2 plane producers outside the split kernels: multi-tensor weight preparation, BN kernels that emit planes, and the
# fused conv -> BN node built on them
# ------------------------------------------------------------------------------------------------
def _h2_decode(buf, rows, ch):
    """h2 split buffer -> (float64 [rows, ch] reconstruction (p0 + p1) * 2^-e, exponent e, raw planes [2, rows, pitch],
    zero tail bytes)."""
    from mit_semseg import _native
    L = _native.lib()
    total = L.semseg_split_h2_bytes(rows, ch)
    assert buf.numel() == total
    plane_bytes = (total - 256 - (256 + 4 * 1024)) // 2
    pitch = plane_bytes // 2 // rows
    raw = buf[:2 * plane_bytes].view(torch.float16).reshape(2, rows, pitch)
    tail = buf[2 * plane_bytes:2 * plane_bytes + 256]
    e = int(buf[2 * plane_bytes + 256:2 * plane_bytes + 260].view(torch.int32).item())
    val = (raw[0, :, :ch].double() + raw[1, :, :ch].double()) * 2.0 ** (-e)
    return val.cpu(), e, raw, tail


@pytest.mark.parametrize('rows,c,ld', [(4096, 96, 96), (300, 150, 160), (77, 24, 40), (8192, 512, 512)], ids=str)
def test_split_h2_with_known_bounds_and_bound_sum(rows, c, ld):
    """semseg_split_h2_bounds: with the exact max|x| as the bound the planes are those of semseg_split_h2 bit for bit; with bounds
    that are loose by 3x (two of them, the larger counts) the planes still reconstruct x to 2^-21 of the bound; semseg_bound_sum
    gives a scalar >= the sum of its terms"""
    import ctypes
    from mit_semseg import ops, _native
    L = _native.lib()
    vp = ctypes.c_void_p
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, ld, generator=g) * 3.0).to(dev())
    st = vp(torch.cuda.current_stream().cuda_stream)
    ref = ops.SCHEMES['h2'].split(x, rows, c, ld)
    exact = x[:, :c].abs().max().reshape(1)

    def split(bounds):
        out = torch.empty(L.semseg_split_h2_bytes(rows, c), dtype=torch.uint8, device=dev())
        bp = (vp * len(bounds))(*[b.data_ptr() for b in bounds])
        _native.check(L.semseg_split_h2_bounds(vp(x.data_ptr()), ld, vp(out.data_ptr()), rows, c, bp, len(bounds), st), 'split_bounds')
        torch.cuda.synchronize()
        return out
    got = split([exact])
    v0, e0, raw0, tail0 = _h2_decode(ref, rows, c)
    v1, e1, raw1, tail1 = _h2_decode(got, rows, c)
    assert e0 == e1 and torch.equal(raw0, raw1) and int(tail1.max()) == 0
    loose = split([exact * 0.5, exact * 3.0])
    v2, e2, _, _ = _h2_decode(loose, rows, c)
    assert e2 in (e0 - 1, e0 - 2)
    err = (v2 - x[:, :c].double().cpu()).abs().max().item()
    assert err <= 2.0 ** -21 * 3.0 * float(exact.item()), err
    terms = [torch.rand(1, generator=g).to(dev()) * 10 ** k for k in range(4)]
    s_out = ops.bound_sum(terms)
    torch.cuda.synchronize()
    want = sum(float(t.item()) for t in terms)
    assert want <= float(s_out.item()) <= want * (1 + 1e-5)
    assert ops.bound_sum(terms[:1]) is terms[0]


@pytest.mark.parametrize('k,c,r', [(64, 3, 3), (64, 64, 3), (150, 512, 1), (512, 1024, 3), (48, 48, 3), (180, 720, 3),
                                   (2048, 1024, 1), (96, 48, 3), (512, 4096, 3)], ids=str)
def test_weights_prepare_h2_matches_split(k, c, r):
    """csrc/weights_prep.hip == semseg_split_h2 of the KRSC weight and of its CRSK transpose, bit for bit (planes incl.
    channel padding, zero tail, exponent word)"""
    from mit_semseg import ops, _native
    L = _native.lib()
    g = torch.Generator().manual_seed(k * 7 + c)
    w = (torch.randn(k, c, r, r, generator=g) * 0.05).to(dev()).contiguous(memory_format=torch.channels_last)
    if r == 1:
        w = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    w2 = (torch.randn(32, 16, 1, 1, generator=g)).to(dev()).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wp = torch.nn.Parameter(w)
    wp2 = torch.nn.Parameter(w2)
    assert ops.prepare_conv_weights([wp2, wp]) == 2          # two tensors in one launch (tile_base bookkeeping)
    krsc, crsk = ops.weight_planes(wp, 'h2')
    assert krsc is not None
    sch = ops.SCHEMES['h2']
    t = r * r
    ref_k = sch.split(ops.krsc(w), k * t, c, c)
    wt = torch.empty((c, r, r, k), device=dev())
    _native.check(L.semseg_weight_krsc_to_crsk(ops._p(ops.krsc(w)), ops._p(wt), k, t, c, ops._st()), 'transpose')
    ref_c = sch.split(wt, c * t, k, k)
    torch.cuda.synchronize()
    for got, ref, rows, ch in ((krsc, ref_k, k * t, c), (crsk, ref_c, c * t, k)):
        gv, ge, graw, gtail = _h2_decode(got, rows, ch)
        rv, re_, rraw, rtail = _h2_decode(ref, rows, ch)
        assert ge == re_
        chp = (ch + 31) // 32 * 32
        assert torch.equal(graw[:, :, :chp].view(torch.int16), rraw[:, :, :chp].view(torch.int16))
        assert int(gtail.max()) == 0
    # the version counter invalidates the planes when the parameter is changed through torch
    with torch.no_grad():
        wp.mul_(2.0)
    assert ops.weight_planes(wp, 'h2') == (None, None)


@pytest.mark.parametrize('n,c,h,w', [(2, 64, 16, 16), (2, 512, 8, 8), (1, 48, 9, 7), (2, 2048, 2, 2), (2, 40, 5, 5), (2, 1024, 32, 32),
                                     (2, 320, 24, 24), (2, 256, 64, 64)], ids=str)
@pytest.mark.parametrize('relu,res', [(False, False), (True, False), (True, True), (False, True)])
def test_bn_h2_forward_kernels(n, c, h, w, relu, res):
    """stats_mm / finalize_mm / apply_h2: y identical to the plain BN kernels, planes reconstruct y to 2^-21, the
    exponent comes from a valid (and, without residual, exact) bound"""
    from mit_semseg import ops, _native
    L = _native.lib()
    d = dev()
    g = torch.Generator().manual_seed(n * 1000 + c)
    P = n * h * w
    z = (torch.randn(P, c, generator=g) * 3 + 1).to(d)
    z[::7] *= 40.0                                             # outliers: exercise the subnormal floor of the low part
    gamma = (torch.rand(c, generator=g) + 0.5).to(d)
    gamma[::3] *= -1                                            # negative scale: min/max swap roles
    beta = torch.randn(c, generator=g).to(d)
    r = (torch.randn(P, c, generator=g) * 2).to(d) if res else None
    rabs = r.abs().max().reshape(1) * 1.5 if res else None     # any upper bound of |res|
    rm, rv = torch.zeros(c, device=d), torch.ones(c, device=d)
    rm2, rv2 = torch.zeros(c, device=d), torch.ones(c, device=d)
    st = ops._st()
    # plain path
    stats0 = torch.empty(2 * c + 1, dtype=torch.float64, device=d)
    ws = torch.empty(max(L.semseg_bn_workspace_bytes(P, c), L.semseg_bn_mm_workspace_bytes(P, c)), dtype=torch.uint8, device=d)
    coef0 = torch.empty(4, c, device=d)
    _native.check(L.semseg_bn_stats(ops._p(z), P, c, ops._p(stats0), ops._p(ws), ws.numel(), st), 'stats')
    _native.check(L.semseg_bn_finalize(ops._p(stats0), c, ops._p(gamma), ops._p(beta), ops._p(rm), ops._p(rv), ops._p(None),
                                       0.1, 1e-5, ops._p(coef0[0]), ops._p(coef0[1]), ops._p(coef0[2]), ops._p(coef0[3]), st), 'fin')
    y0 = torch.empty(P, c, device=d)
    _native.check(L.semseg_bn_apply(ops._p(z), ops._p(coef0[2]), ops._p(coef0[3]), ops._p(r), c, int(relu), ops._p(y0), c, P, c, st), 'apply')
    # mm path
    stats = torch.empty(2 * c + 1, dtype=torch.float64, device=d)
    zmm = torch.empty(2 * c, device=d)
    coef = torch.empty(4, c, device=d)
    absmax = torch.empty(1, device=d)
    yp = torch.full((L.semseg_split_h2_bytes(P, c),), 0x5a, dtype=torch.uint8, device=d)
    y = torch.empty(P, c, device=d)
    _native.check(L.semseg_bn_stats_mm(ops._p(z), P, c, ops._p(stats), ops._p(zmm), ops._p(ws), ws.numel(), st), 'stats_mm')
    _native.check(L.semseg_bn_finalize_mm(ops._p(stats), ops._p(zmm), c, ops._p(gamma), ops._p(beta), ops._p(rm2), ops._p(rv2),
                                          ops._p(None), 0.1, 1e-5, int(relu), ops._p(rabs), ops._p(coef[0]), ops._p(coef[1]),
                                          ops._p(coef[2]), ops._p(coef[3]), ops._p(absmax), ops._p(yp), P, st), 'fin_mm')
    _native.check(L.semseg_bn_apply_h2(ops._p(z), ops._p(coef[2]), ops._p(coef[3]), ops._p(r), c, int(relu), ops._p(y), ops._p(yp),
                                       P, c, ops._p(None), ops._p(None), st), 'apply_h2')
    # single-rank fused form: finish + finalize in one kernel, exponent from the per-block bounds inside apply
    stats_f = torch.empty(2 * c + 1, dtype=torch.float64, device=d)
    zmm_f, coef_f, absmax_f = torch.empty(2 * c, device=d), torch.empty(4, c, device=d), torch.empty(1, device=d)
    rm3, rv3 = torch.zeros(c, device=d), torch.ones(c, device=d)
    nbt = torch.zeros((), dtype=torch.long, device=d)
    bb = torch.empty((c + 15) // 16, dtype=torch.int32, device=d)
    yp_f = torch.full((L.semseg_split_h2_bytes(P, c),), 0x5a, dtype=torch.uint8, device=d)
    y_f = torch.empty(P, c, device=d)
    _native.check(L.semseg_bn_fwd_stats_fused(ops._p(z), P, c, ops._p(stats_f), ops._p(zmm_f), ops._p(gamma), ops._p(beta),
                                              ops._p(rm3), ops._p(rv3), ops._p(nbt), 0.1, 1e-5, int(relu), ops._p(rabs),
                                              ops._p(coef_f[0]), ops._p(coef_f[1]), ops._p(coef_f[2]), ops._p(coef_f[3]),
                                              ops._p(bb), ops._p(ws), ws.numel(), st), 'fwd_stats_fused')
    _native.check(L.semseg_bn_apply_h2(ops._p(z), ops._p(coef_f[2]), ops._p(coef_f[3]), ops._p(r), c, int(relu), ops._p(y_f),
                                       ops._p(yp_f), P, c, ops._p(bb), ops._p(absmax_f), st), 'apply_h2_fused')
    # ... and the form for outputs without planes: the bound itself comes from the finish kernel (max over its blocks)
    stats_b, zmm_b, coef_b = torch.empty_like(stats_f), torch.empty_like(zmm_f), torch.empty_like(coef_f)
    absmax_b = torch.full((1,), 7e30, device=d)
    bb_b = torch.empty_like(bb)
    _native.check(L.semseg_bn_fwd_stats_fused_bound(ops._p(z), P, c, ops._p(stats_b), ops._p(zmm_b), ops._p(gamma), ops._p(beta),
                                                    ops._p(None), ops._p(None), ops._p(None), 0.1, 1e-5, int(relu), ops._p(rabs),
                                                    ops._p(coef_b[0]), ops._p(coef_b[1]), ops._p(coef_b[2]), ops._p(coef_b[3]),
                                                    ops._p(bb_b), ops._p(ws), ws.numel(), st, ops._p(absmax_b)), 'fwd_stats_fused_bound')
    if relu and c % 8 == 0:
        # ... and the ReLU decisions as a bitmask next to the same y / planes
        y_g, yp_g = torch.empty_like(y_f), torch.full_like(yp_f, 0x5a)
        gate_bits = torch.full((P * (c // 8),), 0xff, dtype=torch.uint8, device=d)
        _native.check(L.semseg_bn_apply_h2_gate(ops._p(z), ops._p(coef_f[2]), ops._p(coef_f[3]), ops._p(r), c, 1, ops._p(y_g),
                                                ops._p(yp_g), P, c, ops._p(bb), ops._p(None), st, ops._p(gate_bits)), 'apply_h2_gate')
        torch.cuda.synchronize()
        want = ((y_f > 0).view(P, c // 8, 8).to(torch.int32) << torch.arange(8, device=d, dtype=torch.int32)).sum(-1).to(torch.uint8)
        assert torch.equal(y_g, y_f) and torch.equal(yp_g, yp_f) and torch.equal(gate_bits.view(P, c // 8), want)
    torch.cuda.synchronize()
    assert torch.equal(stats_b, stats_f) and torch.equal(coef_b, coef_f) and torch.equal(bb_b, bb)
    assert absmax_b.item() == absmax_f.item() == absmax.item()
    assert torch.equal(stats_f, stats) and torch.equal(zmm_f, zmm) and int(nbt.item()) == 1
    for a, b in ((coef_f, coef), (rm3, rm2), (rv3, rv2), (y_f, y)):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    vf, ef, rawf, tailf = _h2_decode(yp_f, P, c)
    assert ef == _h2_decode(yp, P, c)[1] and absmax_f.item() >= y_f.abs().max().item()
    assert ((vf - y_f.double().cpu()).abs() <= 2.0 ** -21 * y_f.double().cpu().abs() + 2.0 ** (-25 - ef)).all()
    assert int(tailf.max()) == 0
    assert torch.equal(stats, stats0)
    for a, b in ((coef, coef0), (rm, rm2), (rv, rv2), (y, y0)):     # same formulas, separately compiled kernels
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    assert torch.equal(zmm[:c], z.min(0).values) and torch.equal(zmm[c:], z.max(0).values)
    val, e, raw, tail = _h2_decode(yp, P, c)
    ymax = y.abs().max().item()
    assert absmax.item() >= ymax
    if not res:
        assert absmax.item() == ymax                               # exact without a residual
    assert ymax * 2.0 ** e < 2.0 ** 15 and absmax.item() * 2.0 ** e >= 2.0 ** 14
    yd = y.double().cpu()
    err = (val - yd).abs()
    assert (err <= 2.0 ** -21 * yd.abs() + 2.0 ** (-25 - e)).all(), err.max()
    assert int(tail.max()) == 0
    chp = (c + 31) // 32 * 32
    assert (raw[:, :, c:chp] == 0).all()


@pytest.mark.parametrize('n,c,h,w', [(2, 64, 16, 16), (2, 512, 8, 8), (1, 48, 9, 7), (2, 2048, 2, 2), (2, 1024, 32, 32), (2, 320, 24, 24),
                                     (2, 256, 64, 64)], ids=str)
@pytest.mark.parametrize('relu,dres', [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize('training', [True, False])
def test_bn_h2_backward_kernels(n, c, h, w, relu, dres, training):
    """reduce_mm / bound / bwd_apply_h2: sums and parameter gradients identical to the plain kernels, the planes
    reconstruct the plain fp32 dz to 2^-21, no fp16 overflow"""
    from mit_semseg import ops, _native
    L = _native.lib()
    d = dev()
    g = torch.Generator().manual_seed(n * 1000 + c + 1)
    P = n * h * w
    z = (torch.randn(P, c, generator=g) * 2 + 0.5).to(d)
    dy = (torch.randn(P, c, generator=g) * 1e-3).to(d)
    dy[::5] *= 300.0
    gamma = (torch.rand(c, generator=g) + 0.5).to(d)
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    y = torch.relu((z - mean) * invstd * gamma) if relu else None
    count = torch.tensor([float(P)], dtype=torch.float64, device=d)
    zmm = torch.cat([z.min(0).values, z.max(0).values])
    st = ops._st()
    ws = torch.empty(max(L.semseg_bn_workspace_bytes(P, c), L.semseg_bn_mm_workspace_bytes(P, c)), dtype=torch.uint8, device=d)
    sums0 = torch.empty(2 * c, dtype=torch.float64, device=d)
    dg0, db0 = torch.empty(c, device=d), torch.empty(c, device=d)
    _native.check(L.semseg_bn_bwd_reduce(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), int(relu), P, c,
                                         ops._p(sums0), ops._p(dg0), ops._p(db0), ops._p(ws), ws.numel(), st), 'reduce')
    dz0 = torch.empty(P, c, device=d)
    dres0 = torch.empty(P, c, device=d) if dres else None
    _native.check(L.semseg_bn_bwd_apply(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), ops._p(gamma),
                                        ops._p(sums0), ops._p(count), int(training), int(relu), ops._p(dz0), ops._p(dres0), P, c, st), 'apply')
    sums = torch.empty(2 * c, dtype=torch.float64, device=d)
    gmax = torch.empty(c, device=d)
    dg, db = torch.empty(c, device=d), torch.empty(c, device=d)
    _native.check(L.semseg_bn_bwd_reduce_mm(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), int(relu), P, c,
                                            ops._p(sums), ops._p(gmax), ops._p(dg), ops._p(db), ops._p(ws), ws.numel(), st), 'reduce_mm')
    dzp = torch.full((L.semseg_split_h2_bytes(P, c),), 0x5a, dtype=torch.uint8, device=d)
    dres1 = torch.empty(P, c, device=d) if dres else None
    _native.check(L.semseg_bn_bwd_bound(ops._p(sums), ops._p(count), ops._p(gmax), ops._p(zmm), ops._p(mean), ops._p(invstd),
                                        ops._p(gamma), c, int(training), ops._p(dzp), P, st), 'bound')
    _native.check(L.semseg_bn_bwd_apply_h2(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), ops._p(gamma),
                                           ops._p(sums), ops._p(count), int(training), int(relu), ops._p(dzp), ops._p(dres1), P, c,
                                           ops._p(None), ops._p(None), ops._p(None), st), 'apply_h2')
    # single-rank fused form (+ ReLU gate recomputed from z when there is no residual gradient, i.e. no residual)
    gate = relu and not dres
    gsc = (invstd * gamma) if gate else None
    gsh = (-mean * invstd * gamma) if gate else None
    if gate:       # the gate kernel evaluates fmaf(z, scale, shift): build y from exactly that expression
        y = torch.relu(torch.addcmul(gsh, z, gsc))
    sums_f = torch.empty(2 * c, dtype=torch.float64, device=d)
    dg_f, db_f = torch.empty(c, device=d), torch.empty(c, device=d)
    bb = torch.empty((c + 15) // 16, dtype=torch.int32, device=d)
    dzp_f = torch.full((L.semseg_split_h2_bytes(P, c),), 0x5a, dtype=torch.uint8, device=d)
    dres_f = torch.empty(P, c, device=d) if dres else None
    _native.check(L.semseg_bn_bwd_reduce_fused(ops._p(dy), c, ops._p(None if gate else y), c, ops._p(z), ops._p(mean), ops._p(invstd),
                                               ops._p(gsc), ops._p(gsh), int(relu), P, c, ops._p(count), ops._p(zmm), ops._p(gamma),
                                               int(training), ops._p(sums_f), ops._p(dg_f), ops._p(db_f), ops._p(bb), ops._p(ws),
                                               ws.numel(), st), 'reduce_fused')
    _native.check(L.semseg_bn_bwd_apply_h2(ops._p(dy), c, ops._p(None if gate else y), c, ops._p(z), ops._p(mean), ops._p(invstd),
                                           ops._p(gamma), ops._p(sums_f), ops._p(count), int(training), int(relu), ops._p(dzp_f),
                                           ops._p(dres_f), P, c, ops._p(gsc), ops._p(gsh), ops._p(bb), st), 'apply_h2_fused')
    if relu and not gate and c % 8 == 0:
        # the forward's ReLU bitmask in place of y (y = mask, y_ld = 0; semseg_bn_apply_h2_gate's layout): identical results
        bits = (y > 0).view(P, c // 8, 8).to(torch.int32)
        mask = (bits << torch.arange(8, device=d, dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous()
        sums_m = torch.empty_like(sums_f)
        dg_m, db_m, bb_m = torch.empty_like(dg_f), torch.empty_like(db_f), torch.empty_like(bb)
        dzp_m = torch.full_like(dzp_f, 0x5a)
        dres_m = torch.empty(P, c, device=d) if dres else None
        _native.check(L.semseg_bn_bwd_reduce_fused(ops._p(dy), c, ops._p(mask), 0, ops._p(z), ops._p(mean), ops._p(invstd),
                                                   ops._p(None), ops._p(None), 1, P, c, ops._p(count), ops._p(zmm), ops._p(gamma),
                                                   int(training), ops._p(sums_m), ops._p(dg_m), ops._p(db_m), ops._p(bb_m), ops._p(ws),
                                                   ws.numel(), st), 'reduce_fused_mask')
        _native.check(L.semseg_bn_bwd_apply_h2(ops._p(dy), c, ops._p(mask), 0, ops._p(z), ops._p(mean), ops._p(invstd),
                                               ops._p(gamma), ops._p(sums_m), ops._p(count), int(training), 1, ops._p(dzp_m),
                                               ops._p(dres_m), P, c, ops._p(None), ops._p(None), ops._p(bb_m), st), 'apply_h2_mask')
        torch.cuda.synchronize()
        assert torch.equal(sums_m, sums_f) and torch.equal(dg_m, dg_f) and torch.equal(db_m, db_f) and torch.equal(bb_m, bb)
        assert torch.equal(dzp_m, dzp_f) and (not dres or torch.equal(dres_m, dres_f))
    torch.cuda.synchronize()
    if gate:       # reference for the gated variant: the plain kernels on the y built from the same fmaf
        _native.check(L.semseg_bn_bwd_reduce(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), 1, P, c,
                                             ops._p(sums0), ops._p(dg0), ops._p(db0), ops._p(ws), ws.numel(), st), 'reduce')
        _native.check(L.semseg_bn_bwd_apply(ops._p(dy), c, ops._p(y), c, ops._p(z), ops._p(mean), ops._p(invstd), ops._p(gamma),
                                            ops._p(sums0), ops._p(count), int(training), 1, ops._p(dz0), ops._p(None), P, c, st), 'apply')
        torch.cuda.synchronize()
        dzf_ref = dz0.double().cpu()
        assert torch.equal(sums_f, sums0) and torch.equal(dg_f, dg0) and torch.equal(db_f, db0)
    else:
        dzf_ref = dz0.double().cpu()
        assert torch.equal(sums_f, sums0) and torch.equal(dg_f, dg0) and torch.equal(db_f, db0)
        if dres:
            assert torch.equal(dres_f, dres0)
    vf, ef, rawf, tailf = _h2_decode(dzp_f, P, c)
    fmax = dzf_ref.abs().max().item()
    assert fmax * 2.0 ** ef < 2.0 ** 15 and fmax * 2.0 ** ef >= 2.0 ** 10
    assert ((vf - dzf_ref).abs() <= 2.0 ** -21 * dzf_ref.abs() + 2.0 ** (-25 - ef) + 1e-6 * fmax).all()
    assert int(tailf.max()) == 0
    if gate:
        return     # the unfused comparisons below used the pre-gate y
    assert torch.equal(sums, sums0) and torch.equal(dg, dg0) and torch.equal(db, db0)
    gg = dy * (y > 0) if relu else dy
    assert torch.equal(gmax, gg.abs().max(0).values)
    if dres:
        assert torch.equal(dres1, dres0)
    val, e, raw, tail = _h2_decode(dzp, P, c)
    dzd = dz0.double().cpu()
    dmax = dzd.abs().max().item()
    assert dmax * 2.0 ** e < 2.0 ** 15                        # the bound holds
    assert dmax * 2.0 ** e >= 2.0 ** 10, (dmax, e)            # ... and is not absurdly loose
    err = (val - dzd).abs()
    # 1e-6 * max: the two apply kernels are compiled separately (fma contraction of g - m - xhat*x may differ)
    assert (err <= 2.0 ** -21 * dzd.abs() + 2.0 ** (-25 - e) + 1e-6 * dmax).all(), err.max()
    assert int(tail.max()) == 0


@pytest.mark.parametrize('case', [(2, 64, 24, 24, 128, 3, 1, 1, 1), (2, 256, 16, 16, 256, 3, 1, 2, 2), (2, 128, 17, 19, 64, 1, 2, 0, 1),
                                  (2, 3, 32, 32, 64, 3, 2, 1, 1)], ids=str)
@pytest.mark.parametrize('relu,res', [(True, False), (True, True), (False, False)])
def test_conv_bn_act_fused_vs_float64(case, relu, res, monkeypatch):
    """ops.conv_bn_act (fused node: planes from the BN kernels, dz only as planes, prepared weights) followed by a second
    fused unit that CONSUMES the emitted planes, against float64 conv + batch_norm on the CPU"""
    from mit_semseg import ops
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    n, c, h, w, k, ks, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, c, h, w, generator=g)
    w1 = torch.randn(k, c, ks, ks, generator=g) / (c * ks * ks) ** 0.5
    w2 = torch.randn(k, k, 3, 3, generator=g) / (k * 9) ** 0.5
    g1, b1 = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    g2, b2 = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    oh = (h + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    r = torch.randn(n, k, oh, ow, generator=g) if res else None
    gy = torch.randn(n, k, oh, ow, generator=g)

    def ref():
        ts = [t.double().requires_grad_(True) for t in (x, w1, w2, g1, b1, g2, b2)] + [r.double().requires_grad_(True) if res else None]
        xd, w1d, w2d, g1d, b1d, g2d, b2d, rd = ts
        t = F.batch_norm(F.conv2d(xd, w1d, None, stride, pad, dil), None, None, g1d, b1d, True, 0.1, 1e-5)
        if res:
            t = t + rd
        if relu:
            t = torch.relu(t)
        u = torch.relu(F.batch_norm(F.conv2d(t, w2d, None, 1, 1, 1), None, None, g2d, b2d, True, 0.1, 1e-5))
        u.backward(gy.double())
        return u.detach(), [p.grad for p in ts if p is not None]

    def run():
        ts = [cl(x).requires_grad_(True), torch.nn.Parameter(cl(w1)), torch.nn.Parameter(cl(w2))] + \
             [torch.nn.Parameter(t.to(dev())) for t in (g1, b1, g2, b2)] + [cl(r).requires_grad_(True) if res else None]
        xg, w1g, w2g, g1g, b1g, g2g, b2g, rg = ts
        ops.prepare_conv_weights([w1g, w2g])
        bufs = [torch.zeros(k, device=dev()), torch.ones(k, device=dev()), torch.zeros((), dtype=torch.long, device=dev())]
        if rg is not None:
            ops.attach_absmax(rg, rg.detach().abs().max().reshape(1))
        t = ops.conv_bn_act(xg, w1g, g1g, b1g, bufs[0], bufs[1], bufs[2], residual=rg, stride=stride, padding=pad,
                            dilation=dil, training=True, relu=relu)
        assert (ops.planes_of(t, 'h2', n * oh * ow, k) is not None) == bool(relu)
        u = ops.conv_bn_act(t, w2g, g2g, b2g, torch.zeros(k, device=dev()), torch.ones(k, device=dev()),
                            torch.zeros((), dtype=torch.long, device=dev()), stride=1, padding=1, dilation=1, training=True, relu=True)
        u.backward(cl(gy))
        torch.cuda.synchronize()
        assert int(bufs[2].item()) == 1
        return u.detach(), [p.grad for p in ts if p is not None]

    ur, gr = ref()
    ug, gg = run()
    assert rel_err(ug, ur) < 5e-5, rel_err(ug, ur)
    names = ['x', 'w1', 'w2', 'g1', 'b1', 'g2', 'b2', 'res']
    for nm, a, b in zip(names, gg, gr):
        e = ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
        assert e < 3e-3, (nm, e)      # relative L2; a ReLU gate flipped by a 1e-6 difference moves it by ~1e-3


def test_conv_bn_act_planes_only_output(monkeypatch):
    """conv_bn_act(planes_only=True): the BN output exists as h2 planes only (semseg_bn_apply_h2 with y == NULL) -- the planes are
    bit-identical to those of the full output, the convolution that consumes them gives bit-identical results and gradients, and
    anything that asks for the fp32 values of such a tensor is refused"""
    from mit_semseg import ops, tuner
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(tuner, 'ENABLED', False)
    n, c, h, w, k = 2, 64, 24, 32, 96
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, c, h, w, generator=g).relu()
    w1 = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    w2 = torch.randn(64, k, 1, 1, generator=g) / k ** 0.5
    gy = torch.randn(n, 64, h, w, generator=g)

    def run(planes_only):
        ts = [cl(x).requires_grad_(True), torch.nn.Parameter(cl(w1)), torch.nn.Parameter(cl(w2))]
        xg, w1g, w2g = ts
        ops.prepare_conv_weights([w1g, w2g])
        bn = lambda ch: (torch.ones(ch, device=dev(), requires_grad=True), torch.zeros(ch, device=dev(), requires_grad=True),      # noqa: E731
                         torch.zeros(ch, device=dev()), torch.ones(ch, device=dev()), torch.zeros((), dtype=torch.long, device=dev()))
        b1, b2 = bn(k), bn(64)
        t = ops.conv_bn_act(xg, w1g, *b1, stride=1, padding=1, dilation=1, training=True, relu=True, planes_only=planes_only)
        planes = ops.planes_of(t, 'h2', n * h * w, k)
        assert planes is not None
        assert bool(getattr(t, '_semseg_planes_only', False)) == planes_only
        if planes_only:
            with pytest.raises(RuntimeError):
                ops.as_nhwc(t)
        u = ops.conv_bn_act(t, w2g, *b2, stride=1, padding=0, dilation=1, training=True, relu=True)
        u.backward(cl(gy))
        torch.cuda.synchronize()
        return planes.clone(), u.detach().clone(), [p.grad.clone() for p in ts] + [b1[0].grad.clone(), b1[1].grad.clone()]
    p0, u0, g0 = run(False)
    p1, u1, g1 = run(True)
    used = p0.numel() - (4096 + 256 - 4)          # planes + zero tail + the exponent word (the partial-maxima area of the header is unused here)
    assert torch.equal(p0[:used], p1[:used]) and torch.equal(u0, u1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


# every tile variant of the h2 kernels under a PINNED plan (the tuner only ever runs the fastest one): fwd/dgrad tiles
# 0..5 (register staged 128x128 / 128x64 / 64x64, LDS-DMA 256x128 2-slot / 3-slot ring / 256x256), wgrad tiles 0..3
# (register staged 128 / 64, LDS-DMA 128x128 2-slot / 256x128 3-slot ring / 256x256 2-slot), with and without split-K / split-M
PLAN_CASES = [
    (2, 256, 24, 24, 384, 3, 1, 2, 2),      # dilated 3x3, K not a multiple of 256
    (2, 160, 17, 19, 96, 3, 2, 1, 1),       # stride 2, odd size, C and K with partial 128-blocks
    (1, 512, 16, 16, 1024, 1, 1, 0, 1),     # 1x1
]


@pytest.mark.parametrize('case', PLAN_CASES, ids=str)
@pytest.mark.parametrize('pass_id,tile', [(0, t) for t in range(27)] + [(1, t) for t in range(27)] + [(2, t) for t in list(range(10)) + [11, 12, 13, 14]])
@pytest.mark.parametrize('split', [1, 3])
def test_h2_conv_every_tile_pinned(case, pass_id, tile, split, monkeypatch):
    from mit_semseg import ops, _native, tuner
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(tuner, 'ENABLED', False)
    L = _native.lib()
    n, c, h, w, k, ks, stride, pad, dil = case
    geom = (n, h, w, c, k, ks, ks, stride, pad, dil)
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, ks, ks, generator=g) / (c * ks * ks) ** 0.5
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    if L.semseg_conv2d_h2_set_plan(pass_id, *geom, tile, split) != 0:
        # tiles 22 ... 24 (64-deep k-tiles) take reductions of whole 64-channel chunks only; the library says so at set_plan
        assert tile in (22, 23, 24) and pass_id < 2 and ((((k if pass_id == 1 else c) + 31) // 32) % 2 == 1), (pass_id, tile, case)
        pytest.skip('tile %d does not take this reduction length' % tile)
    try:
        xg, wg = cl(x).requires_grad_(True), cl(wt).requires_grad_(True)
        y = ops.conv2d(xg, wg, None, stride, pad, dil)
        y.backward(cl(gy))
        torch.cuda.synchronize()
    finally:
        L.semseg_conv2d_h2_set_plan(pass_id, *geom, -1, 0)
    got, ref = ((y, yr), (xg.grad, xr.grad), (wg.grad, wr.grad))[pass_id]
    assert rel_err(got, ref) < (REL * 4 if pass_id == 2 else REL), (pass_id, tile, split, rel_err(got, ref))


# wgrad tile 10 (wgrad_taps_kernel: all nine taps of a 3x3 stride-1 pad == dil conv in one block, x staged as a halo) on the
# geometries it accepts -- OW a multiple of 32 -- incl. channel counts that 64 does not divide, several (k, c) tiles, dilation,
# two images (a chunk never crosses an image row), with and without a split over the pixels
@pytest.mark.parametrize('case', [(2, 64, 32, 32, 64, 1), (1, 48, 16, 64, 80, 1), (2, 160, 24, 32, 96, 2), (1, 64, 40, 96, 128, 4),
                                  (2, 3, 32, 64, 64, 1)], ids=str)
@pytest.mark.parametrize('split', [1, 3, 16])
def test_h2_wgrad_all_taps_in_one_block(case, split, monkeypatch):
    from mit_semseg import ops, _native, tuner
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(tuner, 'ENABLED', False)
    L = _native.lib()
    n, c, h, w, k, dil = case
    geom = (n, h, w, c, k, 3, 3, 1, dil, dil)
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    gy = torch.randn(n, k, h, w, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (k, c, 3, 3), gy.double(), stride=1, padding=dil, dilation=dil)
    grads = {}
    for tile in (1, 10):                      # the one-block-per-tap kernel and the all-taps kernel on the same operands
        _native.check(L.semseg_conv2d_h2_set_plan(2, *geom, tile, split), 'set_plan')
        try:
            xg, wg = cl(x).requires_grad_(True), cl(wt).requires_grad_(True)
            ops.conv2d(xg, wg, None, 1, dil, dil).backward(cl(gy))
            torch.cuda.synchronize()
        finally:
            L.semseg_conv2d_h2_set_plan(2, *geom, -1, 0)
        grads[tile] = wg.grad
        assert rel_err(wg.grad, ref) < REL * 4, (tile, split, rel_err(wg.grad, ref))
    # same products, same order inside a chunk; only the chunk boundaries of the split may differ
    assert float((grads[10] - grads[1]).abs().max() / grads[1].abs().max()) < 1e-5
    # geometries the kernel does not take are refused when the plan is pinned (the tuner skips the candidate), not mis-computed:
    # output rows that are not whole 32-pixel chunks, a strided conv, a 1x1 conv
    for bad in ((1, 24, 24, 64, 64, 3, 3, 1, 1, 1), (1, 32, 32, 64, 64, 3, 3, 2, 1, 1), (1, 32, 32, 64, 64, 1, 1, 1, 0, 1)):
        assert L.semseg_conv2d_h2_set_plan(2, *bad, 10, 1) != 0, bad


def test_deferred_wgrad_leaves_computed_weights_alone(monkeypatch):
    """inside ops.defer_wgrad_reduces() only the gradient of a LEAF weight may be finished after backward has returned; a weight
    computed from a parameter (GroupedConv2d's dense expansion) has its gradient read by the next autograd node at once -- it must be
    complete there (found by the SEMSEG_DEPTHWISE_DIRECT=0 golden: garbage of 1e36 in the MobileNetV2 state)"""
    from mit_semseg import ops, tuner
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(tuner, 'ENABLED', False)
    g = torch.Generator().manual_seed(5)
    x = cl(torch.randn(2, 64, 16, 16, generator=g))
    w0 = torch.randn(64, 64, 3, 3, generator=g) / 24
    gy = cl(torch.randn(2, 64, 16, 16, generator=g))
    grads = {}
    for defer in (False, True):
        for leaf in (True, False):
            p = cl(w0.clone()).requires_grad_(True)
            w = p if leaf else p * 2.0
            if defer:
                with ops.defer_wgrad_reduces():
                    ops.conv2d(x, w, None, 1, 1, 1).backward(gy)
            else:
                ops.conv2d(x, w, None, 1, 1, 1).backward(gy)
            torch.cuda.synchronize()
            assert not ops._PENDING_SLABS and not ops._PENDING_WGRADS
            grads[defer, leaf] = p.grad.clone()
    assert torch.equal(grads[True, True], grads[False, True])
    assert torch.equal(grads[True, False], grads[False, False])
    assert rel_err(grads[True, False], 2.0 * grads[True, True].cpu()) < 1e-5      # w = 2 p: products differ by an exact power of two only


def test_deferred_wgrad_only_where_autograd_adopts_the_buffer(monkeypatch):
    """round-4 advice: inside ops.defer_wgrad_reduces() a weight gradient is finished after backward only when the returned buffer
    BECOMES the parameter's .grad.  A standard KCRS-contiguous nn.Parameter (autograd copies the view), a weight used at two sites
    (autograd adds the second gradient to the first) and a parameter with a pre-existing .grad (in-place accumulation) must get the
    same bits as without the deferral -- they used to get a copy of the unreduced buffer."""
    from mit_semseg import ops, tuner
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(tuner, 'ENABLED', False)
    g = torch.Generator().manual_seed(6)
    x = cl(torch.randn(2, 64, 16, 16, generator=g))
    w0 = torch.randn(64, 64, 3, 3, generator=g) / 24
    gy = cl(torch.randn(2, 64, 16, 16, generator=g))

    def run(kind, defer):
        if kind == 'kcrs':
            p = torch.nn.Parameter(w0.clone().cuda())                    # plain contiguous [K, C, R, S]
        else:
            p = cl(w0.clone()).requires_grad_(True)
        if kind == 'preloaded':
            p.grad = torch.zeros_like(p)

        def fb():
            y = ops.conv2d(x, p, None, 1, 1, 1)
            if kind == 'shared':
                y = ops.conv2d(y, p, None, 1, 1, 1)
            y.backward(gy)
        ops._FWD_USES.clear()
        if defer:
            with ops.defer_wgrad_reduces():
                fb()
        else:
            fb()
        torch.cuda.synchronize()
        assert not ops._PENDING_SLABS and not ops._PENDING_WGRADS
        return p.grad.clone()
    for kind in ('krsc', 'kcrs', 'shared', 'preloaded'):
        a, b = run(kind, False), run(kind, True)
        assert torch.isfinite(b).all(), kind
        assert torch.equal(a.contiguous(), b.contiguous()), kind


# (n, h, w, c, k, r, stride, pad, dil, split) -- ragged channel counts, strides, dilation, 1x1 and 3x3, with and without a split
WGRAD_MULTI_GEOMS = [(2, 16, 16, 64, 64, 3, 1, 1, 1, 1), (2, 16, 16, 64, 64, 3, 1, 1, 1, 2), (1, 24, 20, 18, 36, 3, 1, 1, 1, 3),
                     (2, 17, 13, 48, 96, 3, 2, 1, 1, 1), (1, 32, 32, 144, 72, 1, 1, 0, 1, 4), (2, 16, 16, 32, 40, 3, 1, 2, 2, 2),
                     (1, 8, 8, 256, 19, 1, 1, 0, 1, 1), (1, 40, 40, 24, 24, 3, 1, 1, 1, 5)]


@pytest.mark.parametrize('count', [1, 5, 24, 25, 53])
def test_h2_wgrad_many_problems_in_one_launch(count):
    """semseg_conv2d_wgrad_multi_h2: the blocks of up to 24 weight gradients side by side in one launch (packed longest first) leave,
    problem by problem, the SAME slabs (bit for bit) as semseg_conv2d_wgrad_slabs_h2 launched per problem -- across the 24-problem
    launch boundary too;
    a geometry whose plan is not the 64 x 64 register-staged tile is refused before anything runs."""
    import ctypes
    from mit_semseg import ops, _native
    L = _native.lib()
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(count)
    probs, pinned = [], set()
    try:
        for i in range(count):
            n, h, w, c, k, r, stride, pad, dil, split = WGRAD_MULTI_GEOMS[i % len(WGRAD_MULTI_GEOMS)]
            geom = (n, h, w, c, k, r, r, stride, pad, dil)
            if geom not in pinned:
                _native.check(L.semseg_conv2d_h2_set_plan(2, *geom, 1, split), 'set_plan')
                pinned.add(geom)
            assert L.semseg_conv2d_wgrad_tile_h2(*geom) == 1
            oh, ow = ops.conv_out_size(h, r, stride, pad, dil), ops.conv_out_size(w, r, stride, pad, dil)
            x = (torch.randn(n, h, w, c, generator=gen) * (1 + i % 3)).to(dev())
            dy = torch.randn(n, oh, ow, k, generator=gen).to(dev())
            xs = ops.SCHEMES['h2'].split(x, n * h * w, c, c)
            dys = ops.SCHEMES['h2'].split(dy, n * oh * ow, k, k)
            nbytes = L.semseg_conv2d_wgrad_slabs_bytes(*geom)
            want = torch.full((nbytes // 4,), float('nan'), device=dev())
            got = torch.full((nbytes // 4,), float('nan'), device=dev())
            splits = ctypes.c_int(0)
            _native.check(L.semseg_conv2d_wgrad_slabs_h2(vp(xs.data_ptr()), vp(dys.data_ptr()), vp(want.data_ptr()), nbytes,
                                                         ctypes.byref(splits), *geom, st), 'slabs')
            probs.append((geom, xs, dys, want, got, splits.value))
        arr = (_native.WgradProblem * count)()
        for q, (geom, xs, dys, want, got, _) in zip(arr, probs):
            q.xs, q.dys, q.slabs, q.slabs_bytes = xs.data_ptr(), dys.data_ptr(), got.data_ptr(), got.numel() * 4
            q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil = geom
        _native.check(L.semseg_conv2d_wgrad_multi_h2(arr, count, st), 'multi')
        torch.cuda.synchronize()
        for i, (geom, xs, dys, want, got, splits) in enumerate(probs):
            assert arr[i].splits == splits, (i, geom)
            assert not torch.isnan(want).any()
            assert torch.equal(got, want), (i, geom)
        # a plan on another tile: refused, nothing written
        geom = probs[0][0]
        _native.check(L.semseg_conv2d_h2_set_plan(2, *geom, 0, 1), 'set_plan')
        probs[0][4].fill_(7.0)
        assert L.semseg_conv2d_wgrad_multi_h2(arr, count, st) != 0
        torch.cuda.synchronize()
        assert bool((probs[0][4] == 7.0).all())
    finally:
        for geom in pinned:
            L.semseg_conv2d_h2_set_plan(2, *geom, -1, 0)


@pytest.mark.parametrize('case', PLAN_CASES + [(2, 64, 40, 40, 64, 3, 1, 1, 1), (2, 1024, 8, 8, 48, 1, 1, 0, 1)], ids=str)
@pytest.mark.parametrize('tile', list(range(27)))
def test_conv_epilogue_statistics_match_the_sweep(case, tile):
    """semseg_conv2d_fwd_stats_h2 (BN statistics of the conv result gathered per wave row in the GEMM epilogue) +
    semseg_bn_fwd_finish_fused against the statistics sweep over the same result (semseg_bn_fwd_stats_fused), for EVERY tile
    form of the forward kernels under a pinned plan: same z (the conv itself is unchanged), bit-identical min / max, fp64 sums
    equal up to their summation order, identical BN coefficients up to one rounding of the mean; a split-K plan reports 0
    partial rows and gathers nothing."""
    import ctypes
    from mit_semseg import ops, _native
    L = _native.lib()
    vp = ctypes.c_void_p
    P_ = lambda t: vp(t.data_ptr())              # noqa: E731
    n, c, h, w, k, ks, stride, pad, dil = case
    geom = (n, h, w, c, k, ks, ks, stride, pad, dil)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(w, ks, stride, pad, dil)
    M = n * oh * ow
    g = torch.Generator().manual_seed(tile + 31 * k)
    x = (torch.randn(n, h, w, c, generator=g) * 1.7 + 0.3).to(dev())
    wt = (torch.randn(k, ks, ks, c, generator=g) / (c * ks * ks) ** 0.5).to(dev())
    xp = ops.SCHEMES['h2'].split(x, n * h * w, c, c)
    wp = ops.SCHEMES['h2'].split(wt, k * ks * ks, c, c)
    st = vp(torch.cuda.current_stream().cuda_stream)
    gamma, beta = (torch.rand(k, generator=g) + 0.5).to(dev()), torch.randn(k, generator=g).to(dev())

    def outputs():
        return dict(stats=torch.zeros(2 * k + 1, dtype=torch.float64, device=dev()), zmm=torch.zeros(2 * k, device=dev()),
                    rm=torch.zeros(k, device=dev()), rv=torch.ones(k, device=dev()), nbt=torch.zeros((), dtype=torch.int64, device=dev()),
                    coef=torch.zeros(4, k, device=dev()), bb=torch.zeros((k + 15) // 16, dtype=torch.int32, device=dev()),
                    bound=torch.full((1,), 123.0, device=dev()))

    def tail(o):
        return (P_(gamma), P_(beta), P_(o['rm']), P_(o['rv']), P_(o['nbt']), 0.1, 1e-5, 1, vp(0), P_(o['coef'][0]), P_(o['coef'][1]),
                P_(o['coef'][2]), P_(o['coef'][3]), P_(o['bb']))
    for split in (1, 2):
        if L.semseg_conv2d_h2_set_plan(0, *geom, tile, split) != 0:
            assert tile in (22, 23, 24) and ((c + 31) // 32) % 2 == 1, (tile, case)        # 64-deep k-tiles: whole 64-channel chunks only
            pytest.skip('tile %d does not take this reduction length' % tile)
        try:
            conv_ws = torch.empty(max(256, L.semseg_conv2d_h2_workspace_bytes(*geom)), dtype=torch.uint8, device=dev())
            stats_ws = torch.empty(L.semseg_conv2d_fwd_stats_bytes(k), dtype=torch.uint8, device=dev())
            z = torch.empty(M, k, device=dev())
            a, b = outputs(), outputs()
            parts = ctypes.c_int(-1)
            _native.check(L.semseg_conv2d_fwd_stats_h2(P_(xp), P_(wp), P_(z), k, *geom, P_(conv_ws), conv_ws.numel(), P_(stats_ws),
                                                       stats_ws.numel(), P_(a['bound']), ctypes.byref(parts), st), 'conv_stats')
            z2 = torch.empty(M, k, device=dev())
            _native.check(L.semseg_conv2d_fwd_h2(P_(xp), P_(wp), vp(0), P_(z2), k, *geom, P_(conv_ws), conv_ws.numel(), st), 'conv')
            torch.cuda.synchronize()
            assert torch.equal(z, z2)
            bn_ws = torch.empty(L.semseg_bn_mm_workspace_bytes(M, k), dtype=torch.uint8, device=dev())
            _native.check(L.semseg_bn_fwd_stats_fused_bound(P_(z), M, k, P_(b['stats']), P_(b['zmm']), *tail(b), P_(bn_ws),
                                                            bn_ws.numel(), st, P_(b['bound'])), 'sweep')
            torch.cuda.synchronize()
            if split > 1 and L.semseg_conv2d_h2_workspace_bytes(*geom) > 0:
                assert parts.value == 0                # split-K slabs: nothing gathered, the caller sweeps
                continue
            if parts.value == 0:
                continue                               # more wave rows than the epilogue form serves
            assert float(a['bound'].item()) == 0.0     # the epilogue zeroed the bound word for the finish kernel's atomics
            _native.check(L.semseg_bn_fwd_finish_fused(P_(stats_ws), stats_ws.numel(), parts.value, M, k, P_(a['stats']), P_(a['zmm']),
                                                       *tail(a), st, vp(0), P_(a['bound'])), 'finish')
            torch.cuda.synchronize()
            assert torch.equal(a['zmm'], b['zmm'])
            ref = b['stats'].cpu()
            assert float(a['stats'][2 * k].item()) == float(M)
            err = (a['stats'].cpu() - ref).abs() / ref.abs().clamp_min(1e-300)
            assert err[:2 * k].max().item() < 1e-12, err.max().item()
            torch.testing.assert_close(a['coef'], b['coef'], rtol=2e-6, atol=1e-7)
            torch.testing.assert_close(a['rv'], b['rv'], rtol=1e-6, atol=1e-8)
            assert abs(float(a['bound'].item()) - float(b['bound'].item())) <= 1e-5 * abs(float(b['bound'].item()))
            assert int(a['nbt'].item()) == 1
        finally:
            L.semseg_conv2d_h2_set_plan(0, *geom, -1, 0)


def test_inference_weight_planes_follow_sgd_updates(monkeypatch):
    """no_grad forwards build the weight planes once and keep them; the fused SGD kernel (which updates parameters behind
    torch's version counter) must invalidate them"""
    from mit_semseg import ops
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    g = torch.Generator().manual_seed(5)
    x = cl(torch.randn(2, 64, 12, 12, generator=g))
    w = torch.nn.Parameter(cl(torch.randn(96, 64, 3, 3, generator=g) * 0.05))
    with torch.no_grad():
        y0 = ops.conv2d(x, w, None, 1, 1, 1)
        assert ops.weight_planes(w, 'h2')[0] is not None              # cached by the first inference call
        ref0 = F.conv2d(x.cpu().double(), w.detach().cpu().double(), None, 1, 1, 1)
        assert rel_err(y0, ref0) < REL
    grad = cl(torch.randn(96, 64, 3, 3, generator=g))
    buf = torch.empty_like(grad)
    lr = torch.tensor([0.5], device=dev())
    ops.sgd_step([w], [grad], [buf], True, [0.0], lr)                 # raw-pointer update, no version bump
    with torch.no_grad():
        y1 = ops.conv2d(x, w, None, 1, 1, 1)
    torch.cuda.synchronize()
    ref1 = F.conv2d(x.cpu().double(), w.detach().cpu().double(), None, 1, 1, 1)
    assert rel_err(y1, ref1) < REL, rel_err(y1, ref1)
    assert rel_err(y1, ref0) > 1e-2                                    # the weights really moved


@pytest.mark.parametrize('n,c,h,w,sizes', [(2, 64, 64, 64, (1, 2, 3, 6)), (2, 8, 9, 13, (1, 2, 3, 6)), (1, 128, 16, 16, (1, 2, 3, 6)),
                                            (2, 32, 7, 5, (2, 5))], ids=str)
def test_adaptive_avg_pool_multi(n, c, h, w, sizes):
    """all pyramid scales in one pass == nn.AdaptiveAvgPool2d per scale (forward and the summed backward)"""
    from mit_semseg import ops
    g = torch.Generator().manual_seed(n * 100 + c)
    x = torch.randn(n, c, h, w, generator=g)
    gys = [torch.randn(n, c, s, s, generator=g) for s in sizes]
    xr = x.double().requires_grad_(True)
    yr = [F.adaptive_avg_pool2d(xr, s) for s in sizes]
    sum((y * gy.double()).sum() for y, gy in zip(yr, gys)).backward()
    xg = cl(x).requires_grad_(True)
    ys = ops.adaptive_avg_pool_multi(xg, sizes)
    assert len(ys) == len(sizes)
    sum((y * cl(gy)).sum() for y, gy in zip(ys, gys)).backward()
    torch.cuda.synchronize()
    for y, r in zip(ys, yr):
        assert y.shape == r.shape and rel_err(y, r) < 1e-5, rel_err(y, r)
    assert rel_err(xg.grad, xr.grad) < 1e-5, rel_err(xg.grad, xr.grad)


@pytest.mark.parametrize('n,c,h,w,k,dil', [(2, 256, 16, 16, 256, 1), (1, 512, 24, 24, 128, 4), (2, 256, 17, 19, 64, 2),
                                           (1, 320, 7, 9, 36, 1), (2, 1024, 8, 8, 512, 1), (1, 256, 5, 6, 32, 4)], ids=str)
def test_winograd_forward(n, c, h, w, k, dil, monkeypatch):
    """Winograd F(2x2, 3x3) on h2 planes (input transform, batched GEMM over the 16 frequencies, output transform; weights
    transformed by the multi-tensor preparation) against a float64 convolution -- same tolerance as the direct kernels"""
    from mit_semseg import ops, _native
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(ops, 'WINOGRAD', True)
    monkeypatch.setattr(ops, 'WINOGRAD_MIN_C', 64)
    L = _native.lib()
    g = torch.Generator().manual_seed(n * 1000 + c + dil)
    x = torch.randn(n, c, h, w, generator=g).relu() * 1.5
    x.view(-1)[::997] *= 30.0
    wt = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, 1, dil, dil)
    xg = cl(x)
    wp = torch.nn.Parameter(cl(wt))
    ops.prepare_conv_weights([wp])
    u = ops.weight_wino(wp)
    assert u is not None
    z = ops.empty_nhwc(n, k, h, w, xg.device)
    z.fill_(float('nan'))
    bound = xg.abs().max().reshape(1)
    ops._winograd_fwd(L, xg, (bound * 0.5, bound), u, z, (n, h, w, c, k, 3, 3, 1, dil, dil))       # two bounds: max is taken
    torch.cuda.synchronize()
    assert rel_err(z, ref) < REL, rel_err(z, ref)


@pytest.mark.parametrize('n,c,h,w,k,dil', [(2, 256, 16, 16, 256, 1), (1, 512, 24, 24, 128, 4), (2, 256, 17, 19, 64, 2),
                                           (1, 320, 7, 9, 36, 1), (2, 1024, 8, 8, 512, 1), (4, 256, 40, 40, 256, 1)], ids=str)
def test_winograd_wgrad(n, c, h, w, k, dil, monkeypatch):
    """weight gradient in the Winograd domain (dM = A dz A^T from the h2 planes of dz, batched dU[f] = dM[f]^T V[f] on the
    forward's V planes -- with split partials for the last case --, dw = G^T dU G) against a float64 weight gradient"""
    from mit_semseg import ops, _native
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    L = _native.lib()
    g = torch.Generator().manual_seed(n * 1000 + c + dil + 7)
    x = torch.randn(n, c, h, w, generator=g).relu() * 1.5
    x.view(-1)[::997] *= 30.0
    dz = torch.randn(n, k, h, w, generator=g) * 1e-3
    dz.view(-1)[::1013] *= 20.0
    ref = torch.nn.grad.conv2d_weight(x.double(), (k, c, 3, 3), dz.double(), stride=1, padding=dil, dilation=dil)
    geom = (n, h, w, c, k, 3, 3, 1, dil, dil)
    xg, dzg = cl(x), cl(dz)
    wp = torch.nn.Parameter(cl(torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5))
    monkeypatch.setattr(ops, 'WINOGRAD', True)
    monkeypatch.setattr(ops, 'WINOGRAD_MIN_C', 32)
    ops.prepare_conv_weights([wp])
    z = ops.empty_nhwc(n, k, h, w, xg.device)
    v = ops._winograd_fwd(L, xg, (xg.abs().max().reshape(1),), ops.weight_wino(wp), z, geom)
    dzp = ops.SCHEMES['h2'].split(dzg.permute(0, 2, 3, 1), n * h * w, k, k)
    dw = ops._winograd_wgrad(L, v, dzp, geom)
    torch.cuda.synchronize()
    assert dw.shape == ref.shape
    assert rel_err(dw, ref) < REL * 4, rel_err(dw, ref)


@pytest.mark.parametrize('n,c,h,w,k,dil', [(2, 256, 16, 16, 256, 1), (1, 128, 24, 24, 512, 4), (2, 64, 17, 19, 256, 2),
                                           (1, 36, 7, 9, 320, 1), (2, 1024, 8, 8, 512, 1), (2, 4096, 16, 16, 512, 1)], ids=str)
def test_winograd_dgrad(n, c, h, w, k, dil, monkeypatch):
    """data gradient in the Winograd domain (input transform from the h2 planes of dz, the forward's batched GEMM on U' = the
    transformed flipped / transposed weights of the weight preparation, output transform into dx) against a float64 data gradient
    and against the direct h2 data-gradient kernel (same error class)"""
    from mit_semseg import ops, _native
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(ops, 'WINOGRAD', True)
    monkeypatch.setattr(ops, 'WINOGRAD_DGRAD', True)
    monkeypatch.setattr(ops, 'WINOGRAD_MIN_C', 32)
    L = _native.lib()
    g = torch.Generator().manual_seed(n * 1000 + c + dil + 11)
    wt = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    wt.view(-1)[::991] *= 25.0
    dz = torch.randn(n, k, h, w, generator=g) * 1e-3
    dz.view(-1)[::1013] *= 20.0
    ref = torch.nn.grad.conv2d_input((n, c, h, w), wt.double(), dz.double(), stride=1, padding=dil, dilation=dil)
    geom = (n, h, w, c, k, 3, 3, 1, dil, dil)
    wp = torch.nn.Parameter(cl(wt))
    ops.prepare_conv_weights([wp])
    ut = ops.weight_wino_t(wp)
    assert ut is not None
    dzp = ops.SCHEMES['h2'].split(cl(dz).permute(0, 2, 3, 1), n * h * w, k, k)
    dx = ops._winograd_dgrad(L, dzp, ut, geom, form=0)
    torch.cuda.synchronize()
    assert dx.shape == ref.shape
    e_wino = rel_err(dx, ref)
    # the fused kernel (GEMM over the 16 frequencies + output transform in one launch, no fp32 intermediate), both ring depths:
    # same arithmetic per output up to the order in which the 16 frequency terms are added
    monkeypatch.setattr(ops, 'WINOGRAD_FUSED', True)
    tiles = L.semseg_winograd_tiles(n, h, w, dil)
    v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, k), dtype=torch.uint8, device=dzp.device)
    _native.check(L.semseg_winograd_input_planes_h2(ops._p(dzp), ops._p(v), n, h, w, k, dil, ops._st()), 'input_planes')
    for lib_form in range(ops.WINOGRAD_FUSED_FORMS):
        dxf = ops.empty_nhwc(n, c, h, w, dzp.device)
        dxf.fill_(float('nan'))
        _native.check(L.semseg_winograd_gemm_output_h2(ops._p(v), ops._p(ut), ops._p(dxf), c, n, h, w, k, c, dil, lib_form,
                                                       ops._st()), 'gemm_output')
        torch.cuda.synchronize()
        e_f = rel_err(dxf, ref)
        print('   fused form %d: rel err %.2e, max |fused - unfused| / max|dx| %.2e'
              % (lib_form, e_f, float((dxf - dx).abs().max() / dx.abs().max())))
        assert e_f < REL * 4, (lib_form, e_f, e_wino)
        assert float((dxf - dx).abs().max() / dx.abs().max()) < 2e-6
    _, wtp = ops.weight_planes(wp, 'h2')
    dx_direct, _ = ops._split_conv_grads(L, ops.SCHEMES['h2'], 'h2', geom, None, dzp, wp.detach(), wtp, True, False)
    torch.cuda.synchronize()
    e_direct = rel_err(dx_direct, ref)
    print('winograd dgrad %s: rel err %.2e (direct kernel %.2e)' % ((n, c, h, w, k, dil), e_wino, e_direct))
    assert e_wino < REL * 4, (e_wino, e_direct)


def test_conv_bn_act_winograd_wgrad_matches_direct(monkeypatch):
    """the fused node with SEMSEG_WINOGRAD_WGRAD: same weight gradient as the direct wgrad kernels and as float64"""
    from mit_semseg import ops
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    monkeypatch.setattr(ops, 'WINOGRAD', True)
    monkeypatch.setattr(ops, 'WINOGRAD_MIN_C', 64)
    n, c, h, w, k, dil = 2, 192, 20, 24, 128, 2
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(n, c, h, w, generator=g).relu()
    wt = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    ga, be = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    gy = torch.randn(n, k, h, w, generator=g)

    def run(wino_wgrad):
        monkeypatch.setattr(ops, 'WINOGRAD_WGRAD', wino_wgrad)
        xg = cl(x).requires_grad_(True)
        ops.attach_absmax(xg, xg.detach().abs().max().reshape(1))
        wg, gg, bg = torch.nn.Parameter(cl(wt)), torch.nn.Parameter(ga.to(dev())), torch.nn.Parameter(be.to(dev()))
        ops.prepare_conv_weights([wg])
        y = ops.conv_bn_act(xg, wg, gg, bg, torch.zeros(k, device=dev()), torch.ones(k, device=dev()),
                            torch.zeros((), dtype=torch.long, device=dev()), stride=1, padding=dil, dilation=dil,
                            training=True, relu=True)
        y.backward(cl(gy))
        torch.cuda.synchronize()
        return y.detach(), xg.grad, wg.grad

    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = torch.relu(F.batch_norm(F.conv2d(xd, wd, None, 1, dil, dil), None, None, ga.double(), be.double(), True, 0.1, 1e-5))
    yr.backward(gy.double())
    y1, dx1, dw1 = run(True)
    y0, dx0, dw0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert dw1.shape == dw0.shape
    e = ((dw1.double() - dw0.double()).norm() / dw0.double().norm()).item()
    assert e < 1e-5, e
    e = ((dw1.double().cpu() - wd.grad).norm() / wd.grad.norm()).item()
    assert e < 3e-3, e


@pytest.mark.parametrize('n,c,ih,iw,oh,ow', [(1, 150, 8, 10, 35, 45), (2, 150, 64, 64, 64, 64), (1, 7, 5, 3, 9, 12), (1, 64, 6, 6, 13, 7),
                                             (1, 200, 4, 5, 17, 9), (1, 150, 47, 63, 376, 504)], ids=str)
def test_upsample_softmax_fused(n, c, ih, iw, oh, ow):
    """the fused inference head (models.py:480-484 + the multi-scale average of eval.py:66-71): weight * softmax(bilinear(logits))
    written, then a second scale accumulated, against torch's interpolate + softmax in float64"""
    from mit_semseg import ops
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(c * 1000 + oh)
    z1 = torch.randn(n, c, ih, iw, generator=g) * 4
    z2 = torch.randn(n, c, ih + 2, iw + 1, generator=g) * 4

    def ref(z):
        return F.softmax(F.interpolate(z.double(), size=(oh, ow), mode='bilinear', align_corners=False), dim=1)
    single = ops.upsample_softmax(z1.to(d).contiguous(memory_format=torch.channels_last), (oh, ow))
    torch.testing.assert_close(single.cpu().double(), ref(z1), atol=2e-5, rtol=1e-4)     # fp32 exp / sum of 150 terms
    buf = ops.empty_nhwc(n, c, oh, ow, d)
    with ops.head_output(buf, 0.5, False) as h1:
        out = ops.upsample_softmax(z1.to(d), (oh, ow))                      # NCHW-contiguous logits: converted on the way in
    assert out.data_ptr() == buf.data_ptr()
    with ops.head_output(buf, 0.5, True) as h2:
        ops.upsample_softmax(z2.to(d).contiguous(memory_format=torch.channels_last), (oh, ow))
    torch.cuda.synchronize()
    assert h1.used and h2.used
    want = 0.5 * ref(z1) + 0.5 * ref(z2)
    torch.testing.assert_close(buf.cpu().double(), want, atol=2e-5, rtol=1e-4)
    assert torch.equal(buf.cpu().argmax(1), want.float().argmax(1)) or (want.topk(2, dim=1)[0].diff(dim=1).abs().min() < 1e-5)
    assert abs(buf.sum().item() - n * oh * ow) < 1e-3 * n * oh * ow


@pytest.mark.parametrize('rows,c,ld', [(4096, 4096, 4096), (300, 150, 150), (77, 24, 40), (1, 3, 3)], ids=str)
def test_absmax_scalar(rows, c, ld):
    """semseg_absmax (the |x| bound of the evaluation-mode Winograd forward): exact maximum, NaN propagates"""
    from mit_semseg import ops, _native
    d = torch.device('cuda:0')
    x = torch.randn(rows, ld, device=d) * 3
    out = torch.empty(1, device=d)
    ws = ops.workspace(4096, d)
    L = _native.lib()
    _native.check(L.semseg_absmax(ops._p(x), ld, rows, c, ops._p(out), ops._p(ws), ws.numel(), ops._st()), 'absmax')
    assert out.item() == x[:, :c].abs().max().item()
    x[rows // 2, c - 1] = float('nan')
    _native.check(L.semseg_absmax(ops._p(x), ld, rows, c, ops._p(out), ops._p(ws), ws.numel(), ops._st()), 'absmax')
    assert out.item() != out.item()




# ---- side-by-side launches (csrc/batch.h) at operator level --------------------------------------------------------------------
def _scope_stats():
    import ctypes
    from mit_semseg import _native
    out = (ctypes.c_longlong * 4)()
    _native.check(_native.lib().semseg_batch_stats(out), 'batch_stats')
    return dict(scopes=out[0], recorded=out[1], issued=out[2], forced=out[3])


def test_side_by_side_launches_equal_sequential_launches():
    """ops.run_branches inside ops.batch_branches(): SIX branches of different sizes, each a short chain of converted kernels
    (conv -> BN -> ReLU fused node, bilinear up-sampling with accumulate, add + ReLU) plus one UNCONVERTED kernel (max-pool: it forces a
    flush and must still see its producer's result) -- outputs, input gradients and parameter gradients bit-identical to the same
    branches run one after the other; six problems of one kernel leave as a group of four and a group of two (kMaxGroup), the padding
    blocks between problems do nothing."""
    from mit_semseg import ops, tuner
    from mit_semseg.models.layers import Conv2d, BatchNorm2d, ConvBNReLU
    dev = torch.device('cuda:0')
    prev, prev_check = tuner.ENABLED, ops.BATCH_CHECK
    tuner.ENABLED = False
    ops.BATCH_CHECK = True                # every torch operator dispatched inside a scope must be launch-free
    try:
        geoms = [(2, 16, 24, 20), (1, 32, 9, 13), (2, 48, 16, 16), (1, 64, 7, 5), (2, 96, 12, 12), (1, 24, 33, 17)]     # n, c, h, w

        def build(seed):
            torch.manual_seed(seed)
            mods, xs = [], []
            for n, c, h, w in geoms:
                m = ConvBNReLU(Conv2d(c, c + 8, 3, padding=1, bias=False), BatchNorm2d(c + 8)).to(dev).train()
                mods.append(m)
                xs.append(cl(torch.randn(n, c, h, w)).to(dev).requires_grad_(True))
            return mods, xs

        def branch(m, pool):
            def run(x):
                y = m(x)                                               # GEMM, BN finish, BN apply (+ planes)
                if pool:
                    y = ops.max_pool_3x3_s2(y)                         # not a body: flushes what is recorded, then launches
                ya, yb = ops.fork(y)                                   # two consumers: the native fork (autograd's own accumulation
                size = (y.shape[2] * 2, y.shape[3] * 2)                # would be a torch kernel ahead of the recorded launches)
                return ops.add_act(ops.interpolate_bilinear(ya, size), ops.interpolate_bilinear(yb, size), relu=True)
            return ops.Branch(run, [m])

        results = []
        for batched in (False, True):
            mods, xs = build(3)
            # the planes of the inputs and of the weights exist before the scope, as inside TrainStep (the absmax / split / transpose
            # kernels that would make them are not bodies: at the start of every branch they would flush the earlier branches' records)
            for x in xs:
                ops.input_planes(x, 'h2')
            ops.prepare_conv_weights([m._modules['0'].weight for m in mods])
            before = _scope_stats()
            with ops.batch_branches(batched):
                ys = ops.run_branches([branch(m, i == 2) for i, m in enumerate(mods)], xs, side_streams=False)
            loss = sum((y * y).sum() for y in ys)
            with ops.batch_branches(batched):
                loss.backward()
            torch.cuda.synchronize()
            after = _scope_stats()
            if batched:
                assert after['scopes'] - before['scopes'] == 2                                      # forward + backward
                assert after['recorded'] - before['recorded'] > 1.5 * (after['issued'] - before['issued'])
                assert after['forced'] - before['forced'] >= 1                                       # the max-pool of branch 2
            else:
                assert after['scopes'] == before['scopes']
            results.append(([y.detach().clone() for y in ys], [x.grad.clone() for x in xs],
                            [p.grad.clone() for m in mods for p in m.parameters()]))
        for what, a, b in zip(('output', 'input gradient', 'parameter gradient'), results[0], results[1]):
            for i, (t, u) in enumerate(zip(a, b)):
                assert torch.equal(t, u), (what, i, float((t - u).abs().max()), float(t.abs().max()))
        # ... and the check itself: autograd's own accumulation at a tensor with two consumers is a torch kernel inside the scope
        mods, xs = build(4)
        ops.prepare_conv_weights([m._modules['0'].weight for m in mods])
        bad = [ops.Branch(lambda x, m=m: (lambda y: ops.add_act(y, y, relu=True))(m(x)), [m]) for m in mods[:2]]
        with ops.batch_branches(True):
            ys = ops.run_branches(bad, xs[:2], side_streams=False)
            with pytest.raises(RuntimeError, match='side-by-side scope'):
                sum((y * y).sum() for y in ys).backward()
        assert not _native_active()
    finally:
        tuner.ENABLED, ops.BATCH_CHECK = prev, prev_check


def _native_active():
    from mit_semseg import _native
    return bool(_native.lib().semseg_batch_active())
