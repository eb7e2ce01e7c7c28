"""A/B a tuning knob inside ONE process on ONE device (interleaved rounds)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K
from bdvcil_amd._lib import lib
dev = torch.device('cuda:0')
L = lib(); L.bdv_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]; L.bdv_debug_set.restype = None
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
modes = [int(m) for m in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 3]
for (N, H, Cin, Cout, k) in [(256, 16, 256, 256, 3), (256, 14, 256, 256, 3), (256, 14, 1024, 256, 1), (256, 28, 128, 512, 1), (256, 56, 64, 64, 3)]:
    g = K.make_geom(N, H, H, Cin, Cout, k, k, 1, k // 2)
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    res = {m: [[], [], []] for m in modes}
    for rnd in range(3):
        for m in modes:
            L.bdv_debug_set(0, m)
            res[m][0].append(timeit(lambda: K.conv_fprop(x, w, g)))
            res[m][1].append(timeit(lambda: K.conv_dgrad(dy, w, g)))
            res[m][2].append(timeit(lambda: K.conv_wgrad(dy, x, g)))
    print(f'N={N} H={H} {Cin}->{Cout} k{k}: ' + ' | '.join(f'mode{m}: ' + '/'.join(f'{fl / min(t) / 1e9:5.1f}' for t in res[m]) for m in modes))
