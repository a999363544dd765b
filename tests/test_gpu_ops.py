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
