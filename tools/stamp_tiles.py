"""Where a plane-kernel workgroup spends its life: in-kernel s_memtime stamps of the DIAGNOSTIC library
(make -C background-debiased-video-cil_amd/csrc stamps; loaded through BDVCIL_LIB_PATH, the product library has no stamps).
Per site, direction and forced tile: median over workgroups of prologue (start -> first K-step published), K loop, epilogue issue
and store drain in microseconds (ticks -> time through the s_memrealtime pair, 100 MHz), the in-kernel clock, and how the
workgroups' start times spread (rounds).  Dev tool, GPU only:

    BDVCIL_LIB_PATH=background-debiased-video-cil_amd/csrc/libbdvcil_hip_stamps.so python tools/stamp_tiles.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bdvcil_amd import kernels as K
from bdvcil_amd._lib import check, lib

N = int(os.environ.get('N', 256))
dev = torch.device('cuda:0')
# (Cin, Cout, k, stride, H, shift)
SITES = [(256, 1024, 1, 1, 14, 0), (1024, 256, 1, 1, 14, 1), (64, 256, 1, 1, 56, 0), (256, 256, 3, 1, 14, 0), (512, 128, 1, 1, 28, 1),
         (128, 512, 1, 1, 28, 0), (512, 2048, 1, 1, 7, 0)]
TILES = {0: '128x256', 1: '256x128', 2: '256x256', 5: '128x128x4w'}
CAP = 1 << 16

h = lib()
h.bdv_debug_set_stamps.restype = ctypes.c_int
h.bdv_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = torch.zeros(CAP * 8, dtype=torch.int64, device=dev)
check(h.bdv_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()), CAP), 'set_stamps')


def report(tag, nblocks):
    torch.cuda.synchronize()
    q = buf.view(CAP, 8)[:nblocks].cpu().numpy().astype(np.int64)
    q = q[q[:, 4] > 0]                       # workgroups that wrote a result (K-split slices return early)
    if len(q) == 0:
        print(f'  {tag}: no stamps')
        return
    ticks = (q[:, 4] - q[:, 0]).astype(np.float64)
    real = (q[:, 6] - q[:, 5]).astype(np.float64) * 0.01            # us
    clk = np.median(ticks / np.maximum(real, 1e-3))                 # ticks per us = MHz
    us = lambda a, b: np.median((q[:, b] - q[:, a]) / clk)          # noqa: E731
    t0 = (q[:, 5] - q[:, 5].min()) * 0.01
    span = (q[:, 6].max() - q[:, 5].min()) * 0.01
    starts = np.sort(t0)
    print(f'  {tag:24s} wgs {len(q):5d} clock {clk:6.0f} MHz | prologue {us(0, 1):6.2f} kloop {us(1, 2):6.2f} epilogue {us(2, 3):6.2f} '
          f'drain {us(3, 4):6.2f} = wg {np.median(real):6.2f} us | kernel {span:7.1f} us, starts p50 {starts[len(starts) // 2]:6.1f} p99 {starts[int(len(starts) * 0.99)]:6.1f}')


for (Cin, Cout, k, st, H, sh) in SITES:
    g = K.make_geom(N, H, H, Cin, Cout, k, k, st, k // 2, 8, (Cin // 8) if sh else 0)
    x = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    yprev = torch.randn(N, H, H, Cin, device=dev)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (yprev.numel() // 32,), dtype=torch.int32, device=dev)
    stats = (yprev, mask, torch.randn(Cin, device=dev), torch.rand(Cin, device=dev) + 0.5)
    src = torch.randn(N, H, H, Cin, device=dev)
    print(f'site Cin {Cin} Cout {Cout} k{k} s{st} H{H} shift {sh}')
    for kind, ncols in (('fprop', Cout), ('dgrad', Cin)):
        for c, name in TILES.items():
            bn = int(name.split('x')[1])
            if ncols % bn:
                continue
            check(h.bdv_conv_debug_force_tile(c), 'force')
            fn = (lambda: K.conv_fprop(x, w, g, bn_stats=True)) if kind == 'fprop' else \
                ((lambda: K.conv_dgrad(dy, w, g, add_src=src, add_mask_src=mask, bn_stats=stats)) if sh else (lambda: K.conv_dgrad(dy, w, g, bn_stats=stats)))
            for _ in range(3):
                fn()
            buf.zero_()
            fn()
            bm = int(name.split('x')[0])
            M = N * (g.Ho * g.Wo if kind == 'fprop' else H * H)
            report(f'{kind} {name}', min(CAP, ((M + bm - 1) // bm) * (ncols // bn) * 2))
    check(h.bdv_conv_debug_force_tile(-1), 'force')
