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
n, h, w = 1, 8, 8; M = 64
# x one-hot row r (all channels), dy all ones -> dw[k][c] should be 1 everywhere
for r in range(M):
    x = torch.zeros(M, c, device=dev); x[r] = 1.0
    dy = torch.ones(M, k, device=dev)
    dw = wgrad(x, dy, n, h, w, c, k)
    u = torch.unique(dw)
    if not (u.numel() == 1 and u[0].item() == 1.0):
        print('x-onehot row %d: unique %s  nbad %d' % (r, u[:6].tolist(), (dw != 1).sum().item()), flush=True)
# dy = row index value, x = ones: dw[k][c] = sum m = 2016
x = torch.ones(M, c, device=dev)
dy = torch.arange(M, device=dev, dtype=torch.float32).view(M, 1).repeat(1, k).contiguous()
dw = wgrad(x, dy, n, h, w, c, k)
print('dy=m, x=1: unique', torch.unique(dw)[:10].tolist(), 'expected', M*(M-1)//2)
# dy = ones, x = m
dw = wgrad(dy, x, n, h, w, c, k)
print('dy=1, x=m: unique', torch.unique(dw)[:10].tolist())
# determinism + dense random
x = torch.randn(M, c, device=dev); dy = torch.randn(M, k, device=dev)
a = wgrad(x, dy, n, h, w, c, k); b = wgrad(x, dy, n, h, w, c, k)
ref = (dy.double().t() @ x.double())
print('dense random: same twice', torch.equal(a, b), 'err', (a.double()-ref).abs().max().item())
# bf16-exact dense random (values with 8-bit mantissa): only part 0 nonzero
xb = x.bfloat16().float(); dyb = dy.bfloat16().float()
a = wgrad(xb, dyb, n, h, w, c, k); ref = dyb.double().t() @ xb.double()
print('dense bf16-exact: err', (a.double()-ref).abs().max().item())
e = (a.double()-ref).abs()
print('bad entries per k-row (first 8):', (e > 1e-3).sum(1)[:8].tolist(), 'per c-col:', (e > 1e-3).sum(0)[:8].tolist(), 'total bad', (e>1e-3).sum().item())
badk = torch.nonzero((e > 1e-3).sum(1)).flatten().tolist()
print('bad k rows:', badk)
for M2, hh in ((64, 8), (96, 0), (128, 0)):
    pass
# K=C=256 (4 tiles): which (k,c) tiles are bad
c2 = k2 = 256
x = torch.randn(M, c2, device=dev).bfloat16().float(); dy = torch.randn(M, k2, device=dev).bfloat16().float()
a = wgrad(x, dy, n, h, w, c2, k2); ref = dy.double().t() @ x.double()
e = (a.double()-ref).abs() > 1e-3
print('256x256: bad k rows:', torch.nonzero(e.sum(1)).flatten().tolist())
print('256x256: bad c cols count:', (e.sum(0) > 0).sum().item())
