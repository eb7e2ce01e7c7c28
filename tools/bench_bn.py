"""Timing of the BatchNorm kernels at R50 sizes (N = 256 frames).  Dev tool, GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K

dev = torch.device('cuda:0')
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

tot = [0.0, 0.0, 0.0]
# (H, C, count, residual)
for (H, C, cnt, res) in [(112, 64, 1, 0), (56, 64, 6, 0), (56, 256, 4, 1), (56, 128, 1, 0), (28, 128, 7, 0), (28, 512, 5, 1), (28, 256, 1, 0),
                         (14, 256, 11, 0), (14, 1024, 7, 1), (14, 512, 1, 0), (7, 512, 5, 0), (7, 2048, 4, 1)]:
    M = 256 * H * H
    y = torch.randn(M, C, device=dev); dout = torch.randn(M, C, device=dev)
    r = torch.randn(M, C, device=dev) if res else None
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
    mean, invstd, scale, shift = K.bn_train_stats(y, gamma, beta, 1e-5, 0.1, None, None)
    out, mask = K.bn_apply(y, scale, shift, r, True, want_mask=True)
    ta = timeit(lambda: K.bn_apply(y, scale, shift, r, True, out=out, want_mask=True))
    dy = torch.empty_like(y)
    tb = timeit(lambda: K.bn_backward(dout, mask, y, gamma, mean, invstd, True, dy=dy))
    gb = M * C * 4 / 1e9
    pa = (2 + (1 if res else 0)) * gb; pb = 5 * gb
    print(f'{H:4d} {C:5d} x{cnt:<2d} res={res}  apply {ta*1e3:7.1f} us {pa/ta:6.2f} TB/s(x1e-3)   backward {tb*1e3:7.1f} us {pb/tb:6.2f}')
    tot[0] += ta * cnt; tot[1] += tb * cnt
print('total apply %.2f ms  backward %.2f ms' % (tot[0], tot[1]))
