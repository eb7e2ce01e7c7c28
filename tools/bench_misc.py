"""Timing of the small kernels around the conv stack at R50 / B=32 sizes.  Dev tool, GPU only."""
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
    return e0.elapsed_time(e1) / iters * 1e3
a = torch.randn(256, 112, 112, 64, device=dev)
p, idx = K.maxpool_fwd(a)
dp = torch.randn_like(p)
print('maxpool_fwd %.1f us' % timeit(lambda: K.maxpool_fwd(a)))
print('maxpool_bwd %.1f us' % timeit(lambda: K.maxpool_bwd(dp, idx, tuple(a.shape))))
x = torch.randn(256, 2048, device=dev); w = torch.randn(101, 2048, device=dev) * 0.01; b = torch.zeros(101, device=dev)
print('linear_fwd %.1f us' % timeit(lambda: K.linear_fwd(x, w, b)))
do = torch.randn(256, 101, device=dev)
print('linear_bwd %.1f us' % timeit(lambda: K.linear_bwd(do, x, w)))
for (H, C) in [(112, 64), (56, 64), (56, 256), (28, 512), (14, 1024), (7, 2048)]:
    M = 256 * H * H
    part = torch.randn(2, (M + 127) // 128, C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    print('bn_train_finalize M/128=%5d C=%4d: %.1f us' % (part.shape[1], C, timeit(lambda: K.bn_train_finalize(part, M, gamma, beta, 1e-5, 0.1, None, None))))
yy = torch.randn(256, 112, 112, 64, device=dev)
sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev)
def separate():
    a_, m_ = K.bn_apply(yy, sc, sh, None, True, want_mask=True)
    return K.maxpool_fwd(a_)
print('stem tail separate (bn_apply + maxpool_fwd) %.1f us' % timeit(separate))
print('stem tail fused    (bn_relu_maxpool_fwd)    %.1f us' % timeit(lambda: K.bn_relu_maxpool_fwd(yy, sc, sh)))
gam = torch.rand(64, device=dev) + 0.5; bet = torch.zeros(64, device=dev)
mean_, invstd_, sc_, sh_ = K.bn_train_stats(yy, gam, bet, 1e-5, 0.1, None, None)
pp, ii, mm = K.bn_relu_maxpool_fwd(yy, sc_, sh_)
dpp = torch.randn_like(pp)
def sep_bwd():
    da_ = K.maxpool_bwd(dpp, ii, tuple(yy.shape))
    return K.bn_backward(da_, mm, yy, gam, mean_, invstd_, True)
print('stem backward separate (maxpool_bwd + bn_backward) %.1f us' % timeit(sep_bwd))
print('stem backward fused    (bn_backward_maxpool)       %.1f us' % timeit(lambda: K.bn_backward_maxpool(dpp, ii, mm, yy, gam, mean_, invstd_)))
