import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch
from mit_semseg import _native
vp = ctypes.c_void_p
P = lambda t: vp(t.data_ptr())
L = _native.lib(); dev = torch.device('cuda:0')
s = vp(torch.cuda.current_stream().cuda_stream)
def split(t, rows, ch):
    out = torch.empty(L.semseg_split3_bytes(rows, ch), dtype=torch.uint8, device=dev)
    _native.check(L.semseg_split3(P(t), ch, P(out), rows, ch, s), 'split3'); return out
def wgrad(x, dy, n, h, w, c, k):
    M = n*h*w
    ws = torch.empty(max(L.semseg_conv2d_s3_workspace_bytes(n,h,w,c,k,1,1,1,0,1), 1<<20), dtype=torch.uint8, device=dev)
    dw = torch.full((k, c), float('nan'), device=dev)
    xs_, dys_ = split(x, M, c), split(dy, M, k)
    _native.check(L.semseg_conv2d_wgrad_s3(P(xs_), P(dys_), P(dw), n,h,w,c,k,1,1,1,0,1, P(ws), ws.numel(), s), 'w3')
    torch.cuda.synchronize(); return dw
c = k = 128
for hw in (2, 4, 6, 8, 9, 10, 12, 16, 17, 24):
    n, h, w = 1, hw, hw; M = h*w
    x = torch.randn(M, c, device=dev); dy = torch.randn(M, k, device=dev)
    ref = dy.double().t() @ x.double()
    dw = wgrad(x, dy, n, h, w, c, k)
    print('M=%4d err/rms %.2e' % (M, (dw.double()-ref).abs().max().item()/ref.pow(2).mean().sqrt().item()), flush=True)
# one-hot probe at M=256: dy[m*, 0] = 1 ; x[m, c] = m  -> dw[0][c] = m* if the right pixel pairs up
n, h, w = 1, 16, 16; M = 256
x = torch.arange(M, device=dev, dtype=torch.float32).view(M, 1).repeat(1, c).contiguous()
bad = []
for ms in range(M):
    dy = torch.zeros(M, k, device=dev); dy[ms, 0] = 1.0
    dw = wgrad(x, dy, n, h, w, c, k)
    got = dw[0, 0].item()
    if got != ms: bad.append((ms, got))
print('onehot mismatches (m*, got):', bad[:64], len(bad))
