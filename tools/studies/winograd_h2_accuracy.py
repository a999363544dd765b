"""CPU study (no GPU): is Winograd F(2x2, 3x3) on h2 split planes accurate enough for the parity bar (logits within 1e-3,
fp32-class convolution error)?  Emulates exactly what a kernel would compute:
  * input transform  V = B^T d B   in fp32            (d = 4x4 input tile, stride 2)
  * weight transform U = G g G^T   in fp32
  * h2 split of V and U (per-tensor power-of-two scale, fp16 high + fp16 residual), products V0U0 + V0U1 + V1U0 accumulated
    in fp32 over the channels (fp32 matmul on the fp16-representable parts == what the MFMA accumulates, up to summation order)
  * output transform Y = A^T M A   in fp32
and compares with the direct h2 convolution (same split, no Winograd), torch's CPU fp32 convolution, against float64.
Error units as tools/s3_check.py: max|err| / rms(ref) and the componentwise max|err| / sum|a||b|.

    python tools/studies/winograd_h2_accuracy.py            # ~1 min on CPU
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def h2_split(t):
    """per-tensor exponent e with 2^e max|t| in [2^14, 2^15); returns (hi, lo) as fp32 tensors holding fp16 values, and e"""
    m = t.abs().max().item()
    e = 0 if m == 0 else 14 - int(torch.floor(torch.log2(torch.tensor(m))).item())
    s = t * (2.0 ** e)
    hi = s.half().float()
    lo = (s - hi).half().float()
    return hi, lo, e


def h2_matmul(a, b):
    """a [.., M, K] x b [.., K, N] with the 3-product h2 scheme, fp32 accumulation"""
    a0, a1, ea = h2_split(a)
    b0, b1, eb = h2_split(b)
    acc = a1 @ b0
    acc = acc + a0 @ b1
    acc = acc + a0 @ b0
    return acc * (2.0 ** (-(ea + eb)))


def conv_direct_h2(x, w, pad, dil):
    n, c, h, wd = x.shape
    k = w.shape[0]
    cols = F.unfold(x, 3, dilation=dil, padding=pad)                       # [n, c*9, L]
    y = h2_matmul(w.reshape(k, -1)[None], cols)                            # [n, k, L]
    return y.reshape(n, k, h, wd)


def conv_winograd_h2(x, w, pad, dil):
    """stride-1 3x3 (dilated: the taps form a dilation-strided 3x3, handled by subsampling into dil*dil phase images)"""
    n, c, h, wd = x.shape
    k = w.shape[0]
    out = torch.zeros(n, k, h, wd)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)                          # [k, c, 4, 4]
    for ph in range(dil):
        for pw in range(dil):
            xs = x[:, :, ph::dil, pw::dil]                                  # phase image: plain 3x3, pad 1
            hs, ws = xs.shape[2:]
            hp, wp = (hs + 1) // 2 * 2, (ws + 1) // 2 * 2
            xp = F.pad(xs, (1, 1 + wp - ws, 1, 1 + hp - hs))
            tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [n, c, th, tw, 4, 4]
            V = torch.einsum('ij,nctujl,ml->nctuim', BT, tiles, BT)         # [n, c, th, tw, 4, 4]
            th, tw = V.shape[2:4]
            Vm = V.permute(4, 5, 0, 2, 3, 1).reshape(16, n * th * tw, c)    # [16, tiles, c]
            Um = U.permute(2, 3, 1, 0).reshape(16, c, k)                    # [16, c, k]
            M = h2_matmul(Vm, Um)                                           # ONE scale per operand tensor, as the planes have
            M = M.reshape(4, 4, n, th, tw, k)
            Y = torch.einsum('ij,jlnthk,ml->nkthim', AT, M, AT)             # [n, k, th, tw, 2, 2]
            Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, hp, wp)[:, :, :hs, :ws]
            out[:, :, ph::dil, pw::dil] = Y
    return out


def report(name, got, ref, mag):
    e = (got.double() - ref).abs()
    print('   %-18s max|err|/rms %.2e   componentwise %.2e' % (name, e.max().item() / ref.pow(2).mean().sqrt().item(),
                                                              (e / (mag + 1e-300)).max().item()))


def main():
    cases = [('conv_last-like  C=4096 K=128 16x16', 1, 4096, 16, 16, 128, 1), ('layer4 d4-like  C=512 K=128 24x24', 1, 512, 24, 24, 128, 4),
             ('layer3 d2-like  C=256 K=128 20x20', 2, 256, 20, 20, 128, 2), ('stem-like       C=64  K=64 40x40', 1, 64, 40, 40, 64, 1)]
    for name, n, c, h, w, k, dil in cases:
        x = torch.randn(n, c, h, w).relu() * 1.5
        x.view(-1)[::9973] *= 50.0                                          # ReLU'd activations with outliers
        wt = torch.randn(k, c, 3, 3) * (2.0 / (c * 9)) ** 0.5
        ref = F.conv2d(x.double(), wt.double(), None, 1, dil, dil)
        mag = F.conv2d(x.double().abs(), wt.double().abs(), None, 1, dil, dil)
        print(name)
        report('torch cpu fp32', F.conv2d(x, wt, None, 1, dil, dil), ref, mag)
        report('direct h2', conv_direct_h2(x, wt, dil, dil), ref, mag)
        report('winograd fp32', F.conv2d(x, wt, None, 1, dil, dil) * 0 + conv_winograd_fp32(x, wt, dil), ref, mag)
        report('winograd h2', conv_winograd_h2(x, wt, dil, dil), ref, mag)


def conv_winograd_fp32(x, w, dil):
    global h2_matmul
    saved = h2_matmul
    h2_matmul = lambda a, b: a @ b                                          # noqa: E731  (plain fp32 products)
    try:
        return conv_winograd_h2(x, w, dil, dil)
    finally:
        h2_matmul = saved


if __name__ == '__main__':
    main()
