"""VERDICT round-1 item 7(i), measured: fold bn_apply(+ReLU) of a unit into the A-tile loader of the 1x1 conv that consumes it.

For every conv3 site of TSM-R50 (the 1x1 consumers of conv2's BatchNorm + ReLU), N = 256 frames, one process:
  (a) the product path: bdv_bn_apply (writes the activation and the 1-bit mask) + the 1x1 fprop with fused statistics,
      with the planner's kernel choice and with the 128 x 256 tile forced (the tile the experiment kernel has);
  (b) bdv_conv_fprop_pl(pre_scale, pre_shift) on the raw conv output (no apply pass, no activation, no mask written).
Forward only; the whole-step effect (weight gradient with the same loader, ReLU signs derived in the backward kernels) is
tools/ab_step.py's 'apply passes of conv1 / conv2' variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bdvcil_amd import kernels as K
from bdvcil_amd._lib import check, lib
from tools.bench_conv import timeit

dev = torch.device('cuda:0')
N = 256
SITES = [(64, 256, 56, 3), (128, 512, 28, 4), (256, 1024, 14, 6), (512, 2048, 7, 3)]      # Cin, Cout, H, blocks
tot = {'apply': 0.0, 'conv_auto': 0.0, 'conv_128x256': 0.0, 'prebn': 0.0}
print(f'{"site":24s} {"bn_apply":>9s} {"fprop auto":>11s} {"fprop 128x256":>14s} {"prebn fprop":>12s}   apply+auto  vs  prebn')
for Cin, Cout, H, cnt in SITES:
    g = K.make_geom(N, H, H, Cin, Cout, 1, 1, 1, 0)
    y2 = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    scale = torch.rand(Cin, device=dev) + 0.5
    shift = torch.randn(Cin, device=dev) * 0.2
    out2, mask = K.bn_apply(y2, scale, shift, None, True, want_mask=True)
    ref, pref = K.conv_fprop(out2, w, g, bn_stats=True)
    got, pgot = K.conv_fprop(y2, w, g, bn_stats=True, pre_bn=(scale, shift))
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    serr = (pgot.double().sum(1) - pref.double().sum(1)).abs().max().item() / pref.double().sum(1).abs().max().item()
    assert err <= 2e-6 and serr <= 1e-5, (err, serr)
    t_apply = timeit(lambda: K.bn_apply(y2, scale, shift, None, True, want_mask=True))
    t_auto = timeit(lambda: K.conv_fprop(out2, w, g, bn_stats=True))
    check(lib().bdv_conv_debug_force_tile(0), 'force')
    t_f0 = timeit(lambda: K.conv_fprop(out2, w, g, bn_stats=True))
    check(lib().bdv_conv_debug_force_tile(-1), 'force')
    t_pre = timeit(lambda: K.conv_fprop(y2, w, g, bn_stats=True, pre_bn=(scale, shift)))
    print(f'({Cin:4d},{Cout:5d}) @{H:3d} x{cnt}   {t_apply:9.3f} {t_auto:11.3f} {t_f0:14.3f} {t_pre:12.3f}   {t_apply + t_auto:8.3f}   vs {t_pre:7.3f}   (max err {err:.1e})', flush=True)
    for k, t in (('apply', t_apply), ('conv_auto', t_auto), ('conv_128x256', t_f0), ('prebn', t_pre)):
        tot[k] += t * cnt
print(f'per step (16 blocks): bn_apply {tot["apply"]:.3f} ms + fprop {tot["conv_auto"]:.3f} ms = {tot["apply"] + tot["conv_auto"]:.3f} ms '
      f'(fprop with 128x256 forced: {tot["conv_128x256"]:.3f});  pre-BN fprop alone {tot["prebn"]:.3f} ms')
