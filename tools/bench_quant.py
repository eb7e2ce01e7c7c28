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
for (N, H, Cin, Cout, k) in [(256, 14, 256, 256, 3), (256, 16, 256, 256, 3), (192, 16, 256, 256, 3), (128, 16, 256, 256, 3), (256, 16, 256, 512, 3), (256, 16, 256, 1024, 1), (256, 16, 1024, 256, 1), (256, 32, 512, 128, 1), (256, 32, 128, 512, 1), (256, 16, 256, 256, 1), (256, 16, 2048, 256, 1)]:
    g = K.make_geom(N, H, H, Cin, Cout, k, k, 1, k // 2)
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    tf = timeit(lambda: K.conv_fprop(x, w, g)); td = timeit(lambda: K.conv_dgrad(dy, w, g)); tw = timeit(lambda: K.conv_wgrad(dy, x, g))
    MT = (N * H * H + 127) // 128
    print(f'N={N} H={H} {Cin}->{Cout} k{k}: fprop blocks {MT * (Cout // 128)} ({MT * (Cout // 128) / 256:.2f}/CU) {fl / tf / 1e9:6.1f} TF | dgrad blocks {MT * (Cin // 128)} {fl / td / 1e9:6.1f} TF | wgrad {fl / tw / 1e9:6.1f} TF')
