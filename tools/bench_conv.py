"""Per-shape timing of the conv kernels at the TSM-R50 sites (N = 256 frames).  Dev tool, GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K

N = int(os.environ.get('N', 256))
# (Cin, Cout, k, stride, Hin, count, shift)
SHAPES = [
    (4, 64, 7, 2, 224, 1, 0),
    (64, 64, 1, 1, 56, 1, 1), (64, 64, 3, 1, 56, 3, 0), (64, 256, 1, 1, 56, 4, 0), (256, 64, 1, 1, 56, 2, 1),
    (256, 128, 1, 1, 56, 1, 1), (128, 128, 3, 2, 56, 1, 0), (128, 512, 1, 1, 28, 4, 0), (256, 512, 1, 2, 56, 1, 0),
    (512, 128, 1, 1, 28, 3, 1), (128, 128, 3, 1, 28, 3, 0),
    (512, 256, 1, 1, 28, 1, 1), (256, 256, 3, 2, 28, 1, 0), (256, 1024, 1, 1, 14, 6, 0), (512, 1024, 1, 2, 28, 1, 0),
    (1024, 256, 1, 1, 14, 5, 1), (256, 256, 3, 1, 14, 5, 0),
    (1024, 512, 1, 1, 14, 1, 1), (512, 512, 3, 2, 14, 1, 0), (512, 2048, 1, 1, 7, 3, 0), (1024, 2048, 1, 2, 14, 1, 0),
    (2048, 512, 1, 1, 7, 2, 1), (512, 512, 3, 1, 7, 2, 0),
]

def timeit(fn, iters=7):
    """Median of individually timed calls (a mean is thrown off by the occasional allocator stall of a first-size slab)."""
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    tot = {'fprop': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    totf = 0.0
    print(f'{"shape":34s} {"GF":>7s} | {"fprop ms":>8s} {"TF":>6s} | {"dgrad ms":>8s} {"TF":>6s} | {"wgrad ms":>8s} {"TF":>6s}')
    for (Cin, Cout, k, st, H, cnt, sh) in SHAPES:
        pad = k // 2
        g = K.make_geom(N, H, H, Cin, Cout, k, k, st, pad, 8, (Cin // 8) if sh else 0)
        x = torch.randn(N, H, H, Cin, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
        dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * k * k * (3 if Cin == 4 else Cin)
        tf = timeit(lambda: K.conv_fprop(x, w, g))
        td = timeit(lambda: K.conv_dgrad(dy, w, g)) if Cin % 64 == 0 else 0.0
        if K.WGRAD_X3:   # the experimental main kernel + its reduction, as a stage's backward issues them
            def _w():
                slab, dw = K.conv_wgrad_partial(dy, x, g)
                K.wgrad_reduce_batched([(slab, dw)])
            tw = timeit(_w)
        else:
            tw = timeit(lambda: K.conv_wgrad(dy, x, g))
        print(f'{str((Cin, Cout, k, st, H)):28s} x{cnt:<3d} {flops/1e9:7.1f} | {tf:8.3f} {flops/tf/1e9:6.1f} | {td:8.3f} {(flops/td/1e9 if td else 0):6.1f} | {tw:8.3f} {flops/tw/1e9:6.1f}')
        tot['fprop'] += tf * cnt; tot['dgrad'] += td * cnt; tot['wgrad'] += tw * cnt
        totf += flops * cnt
    print('total ms  fprop %.2f  dgrad %.2f  wgrad %.2f  sum %.2f ; conv GFLOP fwd %.1f -> avg TF/s (3x flops / sum) %.1f'
          % (tot['fprop'], tot['dgrad'], tot['wgrad'], sum(tot.values()), totf / 1e9, 3 * totf / sum(tot.values()) / 1e9))
