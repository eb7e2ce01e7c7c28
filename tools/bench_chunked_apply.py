"""Would the forward BatchNorm apply pass hide under the conv that consumes it?  Block-output apply (with residual) followed by the
next block's conv1 (1x1 + temporal shift, fused statistics) at the layer-1 / layer-2 shapes of TSM-R50, batch 32: the two kernels
back to back on one stream, against two clip chunks -- apply(c0); conv(c0) beside apply(c1) on a second stream; conv(c1).
Dev tool, GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K

dev = torch.device('cuda:0')
side = torch.cuda.Stream()


def timeit(fn, iters=9):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for (N, H, C, Cout, k) in [(256, 56, 256, 64, 1), (256, 28, 512, 128, 1), (256, 56, 64, 64, 3), (256, 28, 128, 128, 3), (256, 14, 1024, 256, 1)]:
    fold = C // 8 if k == 1 else 0
    y = torch.randn(N, H, H, C, device=dev)
    res = torch.randn(N, H, H, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    w = torch.randn(Cout, k, k, C, device=dev) * 0.05
    g = K.make_geom(N, H, H, C, Cout, k, k, 1, k // 2, 8, fold)
    gh = K.make_geom(N // 2, H, H, C, Cout, k, k, 1, k // 2, 8, fold)
    a = torch.empty_like(y)
    yo = torch.empty(N, H, H, Cout, device=dev)
    h = N // 2

    def seq():
        K.bn_apply(y, sc, sh, res, True, out=a)
        K.conv_fprop(a, w, g, out=yo, bn_stats=True)

    def chunked():
        main = torch.cuda.current_stream()
        K.bn_apply(y[:h], sc, sh, res[:h], True, out=a[:h])
        ev0 = torch.cuda.Event(); ev0.record(main)
        side.wait_event(ev0)                      # (the finalize that precedes the apply in the real step)
        with torch.cuda.stream(side):
            K.bn_apply(y[h:], sc, sh, res[h:], True, out=a[h:])
            ev1 = torch.cuda.Event(); ev1.record(side)
        K.conv_fprop(a[:h], w, gh, out=yo[:h], bn_stats=True)
        main.wait_event(ev1)
        K.conv_fprop(a[h:], w, gh, out=yo[h:], bn_stats=True)

    t_a = timeit(lambda: K.bn_apply(y, sc, sh, res, True, out=a))
    t_c = timeit(lambda: K.conv_fprop(a, w, g, out=yo, bn_stats=True))
    print(f'{C:5d}@{H:<3d} -> {Cout:4d} k{k}: apply {t_a:.3f} conv {t_c:.3f} back to back {timeit(seq):.3f} chunked {timeit(chunked):.3f} ms', flush=True)
