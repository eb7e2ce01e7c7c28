"""Per-site timing of every kernel choice for fprop / dgrad at the TSM-R50 sites (N = 256 frames), in ONE process (timings
taken on different boxes differ by several per cent): the round-1 bf16-piece kernels (operands split in the K loop, two
workgroups per CU) and the 8-wave kernels on pre-split weight planes with each tile configuration forced in turn.
Prints one row per site and direction with the time of every choice and the best one.  FUSED=1 times the calls as the
training step makes them: fprop with the BatchNorm statistics in its epilogue; dgrad of a block's conv1 (the shifted sites) with
the identity-branch gradient + ReLU mask added and the BatchNorm-backward statistics taken, the other stride-1 dgrads with the
statistics.  Dev tool, GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bdvcil_amd import kernels as K
from bdvcil_amd._lib import check, lib
from tools.bench_conv import SHAPES, timeit  # noqa: E402

N = int(os.environ.get('N', 256))
K.set_conv_arith(os.environ.get('ARITH', 'bf16x3'))      # 'bf16x2': the r1 columns then still are three-piece kernels
FUSED = os.environ.get('FUSED', '0') == '1'
dev = torch.device('cuda:0')
CFG = {0: '128x256', 1: '256x128', 2: '256x256', 3: '256x64', 4: '64x128', 5: '128x128'}


def applicable(c, ncols):
    bn = {0: 256, 1: 128, 2: 256, 3: 64, 4: 128, 5: 128}[c]
    return ncols % bn == 0


print(f'{"site":30s} dir    {"r1 x3":>8s} {"r1 planes":>9s} ' + ' '.join(f'{CFG[c]:>8s}' for c in range(6)) + '   best')
tot = {}
for (Cin, Cout, k, st, H, cnt, sh) in SHAPES:
    if Cin % 32 != 0:
        continue
    g = K.make_geom(N, H, H, Cin, Cout, k, k, st, k // 2, 8, (Cin // 8) if sh else 0)
    x = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    fp, dg = (lambda: K.conv_fprop(x, w, g)), (lambda: K.conv_dgrad(dy, w, g))
    if FUSED:
        fp = lambda: K.conv_fprop(x, w, g, bn_stats=True)  # noqa: E731
        if st == 1:
            yprev = torch.randn(N, H, H, Cin, device=dev)
            mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (yprev.numel() // 32,), dtype=torch.int32, device=dev)
            stats = (yprev, mask, torch.randn(Cin, device=dev), torch.rand(Cin, device=dev) + 0.5)
            if sh:
                src = torch.randn(N, H, H, Cin, device=dev)
                dg = lambda: K.conv_dgrad(dy, w, g, add_src=src, add_mask_src=mask, bn_stats=stats)  # noqa: E731
            else:
                dg = lambda: K.conv_dgrad(dy, w, g, bn_stats=stats)  # noqa: E731
    for name, ncols, fn in (('fprop', Cout, fp), ('dgrad', Cin, dg)):
        K.USE_PL = False
        t_old = timeit(fn)
        K.USE_PL = True
        check(lib().bdv_conv_debug_force_tile(6), 'force')      # two-workgroup kernels with the weights from the planes
        t_r1p = timeit(fn)
        ts = {}
        for c in range(6):
            if not applicable(c, ncols) or (c == 4 and (name != 'dgrad' or st != 1)):
                continue
            check(lib().bdv_conv_debug_force_tile(c), 'force')
            ts[c] = timeit(fn)
        check(lib().bdv_conv_debug_force_tile(-1), 'force')
        t_auto = timeit(fn)
        best = min([('r1', t_old), ('r1p', t_r1p)] + [(CFG[c], t) for c, t in ts.items()], key=lambda q: q[1])
        row = f'{str((Cin, Cout, k, st, H)):26s} x{cnt:<2d} {name}  {t_old:8.3f} {t_r1p:9.3f} ' + ' '.join(
            f'{ts[c]:8.3f}' if c in ts else f'{"-":>8s}' for c in range(6)) + f'   {best[0]:8s} auto {t_auto:.3f}'
        print(row, flush=True)
        for key, t in [('r1', t_old), ('r1p', t_r1p), ('auto', t_auto), ('best', best[1])]:
            tot[(name, key)] = tot.get((name, key), 0.0) + t * cnt
for name in ('fprop', 'dgrad'):
    print(f'total {name}: r1 x3 {tot[(name, "r1")]:.2f} ms, r1 on planes {tot[(name, "r1p")]:.2f} ms, planner {tot[(name, "auto")]:.2f} ms, '
          f'best per site {tot[(name, "best")]:.2f} ms')
