"""Can an HBM-bound BatchNorm kernel run under an MFMA-bound conv kernel on a second stream?  Dev tool, GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K

dev = torch.device('cuda:0')
N = 256
def conv_case(Cin, Cout, k, H):
    g = K.make_geom(N, H, H, Cin, Cout, k, k, 1, k // 2)
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    return g, x, w, dy

g, x, w, dy = conv_case(256, 256, 3, 14)
M, C = 256 * 28 * 28, 512
y = torch.randn(M, C, device=dev); dout = torch.randn(M, C, device=dev)
gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
mean, invstd, scale, shift = K.bn_train_stats(y, gamma, beta, 1e-5, 0.1, None, None)
out, mask = K.bn_apply(y, scale, shift, None, True, want_mask=True)
dyb = torch.empty_like(y)
side = torch.cuda.Stream()
dw = torch.empty(256, 3, 3, 256, device=dev)

def bn_work(n):
    for _ in range(n):
        K.bn_backward(dout, mask, y, gamma, mean, invstd, True, dy=dyb)

def conv_work(kind, n, tag):
    for _ in range(n):
        if kind == 'wgrad': K.conv_wgrad(dy, x, g, dw=dw, ws_tag=tag)
        elif kind == 'dgrad': K.conv_dgrad(dy, w, g)
        else: K.conv_fprop(x, w, g)

def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

for kind in ('wgrad', 'dgrad', 'fprop'):
    conv_work(kind, 2, 'side'); bn_work(2); torch.cuda.synchronize()
    nc, nb = 20, 20
    tc = wall(lambda: conv_work(kind, nc, 'side'))
    tb = wall(lambda: bn_work(nb))
    def both():
        with torch.cuda.stream(side):
            conv_work(kind, nc, 'side')
        bn_work(nb)
    tt = wall(both)
    print(f'{kind}: conv alone {tc:.2f} ms, bn alone {tb:.2f} ms, sum {tc+tb:.2f}, concurrent {tt:.2f} ms')
