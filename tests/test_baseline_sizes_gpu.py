"""Parity at the BASELINE sizes of configs 3, 4 and 5 (BASELINE.json ``configs[2..4]``), whose planner branches (whole rounds,
K-split of remainder tiles, XCD remap, parity classes) depend on the launch size and are not reached by the small-shape tests.

* config 5 -- TSM-R50, bf16 storage, batch 64 (N = 512 frames): every non-stem TSM-R50 conv site, bf16-STORAGE kernels against
  the fp32-storage ``bf16x1`` kernels on the same values (GPU vs GPU; the bar of test_bf16_storage_gpu.py: one bf16 ulp of the
  result + 2e-5 of scale, > 98 % of the elements exactly the rounding), and for eight sites across the planner's branches fprop
  and dgrad against torch CPU fp32 on the bf16-rounded operands (the arithmetic's definition: operands rounded to bf16, exact
  products, fp32 accumulation).  The reference trains in precision 32 (libs/cil/cil.py:744-756): parity unpinned, own bars.
* config 4 -- I3D-R50, 16 clips x 32 frames x 224^2 (configs/_base_/models/i3d_r50.py:1-27): every distinct conv geometry of
  the network at that size -- the seven 3x1x1 temporal sites (kx1 convolution on the [B][T][H*W][C] view), the eighteen
  per-frame sites at their frame counts (N = 128 in layer1, 64 after pool2) and the 5x7x7 / (2,2,2) stem -- fprop / dgrad /
  wgrad against torch CPU ``conv3d`` / ``conv2d`` + autograd at 2e-5 of scale (weight gradients: 1e-4 against the fp32 CPU
  result, 256 sampled entries at 2e-5 against fp64 dot products, as in test_conv_sites_gpu.py).
* config 3 -- one CIL task-1 step at B = 32 (uint8 background-mix front-end, frozen teacher, five feature-KD terms, LSC head +
  LSCLoss; libs/cil/cil.py:512-556): properties any correct implementation has -- the KD terms equal an independent
  torch reduction over the hooked tensors, the total is loss_cls + scale * sum(w_m * mse_m), every gradient is finite, the
  step is bit-reproducible, backward is linear in the loss gradient, and the teacher's eval forward treats clips independently.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.tsm_oracle import temporal_shift
from test_conv_sites_gpu import SITES, _err, check_site      # (tests/ is on sys.path: pytest rootdir import)

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


# ---------------------------------------------------------------------------------------------------------------------
# config 5: N = 512 frames, bf16 storage
# ---------------------------------------------------------------------------------------------------------------------
N5 = 512
# checked against torch CPU as well: 64-wide tiles, a 3x3 stride-1 site per stage, a stride-2 1x1 (parity classes), shifted
# conv1 sites with K = 1024 / 2048, a wide-output 1x1, the layer-4 3x3 (98 row tiles of 256)
CPU_CHECKED = {(64, 64, 3, 1, 56, 0), (128, 128, 3, 1, 28, 0), (256, 512, 1, 2, 56, 0), (1024, 256, 1, 1, 14, 1),
               (256, 256, 3, 1, 14, 0), (256, 1024, 1, 1, 14, 0), (2048, 512, 1, 1, 7, 1), (512, 512, 3, 1, 7, 0)}


def _bound_ok(out16, ref32, frac=None):
    """|out - ref| <= one bf16 ulp of ref + 2e-5 of scale (summation-order noise); optionally: most elements ARE the rounding"""
    scale = ref32.abs().max().item() + 1e-12
    err = (out16.float() - ref32).abs()
    bound = ref32.abs() * 2.0 ** -8 + 2e-5 * scale
    worst = (err - bound).max().item()
    assert worst <= 0, worst
    if frac is not None:
        assert (out16 == ref32.to(BF)).float().mean().item() > frac


@pytest.mark.parametrize('site', [s for s in SITES if s[0] != 3], ids=lambda s: 'x'.join(map(str, s)))
def test_config5_sites_n512_bf16_storage(site, dev):
    from bdvcil_amd import kernels as K
    Cin, Cout, k, st, H, shift = site
    pad, fold, T = k // 2, (Cin // 8 if shift else 0), (8 if shift else 1)
    Ho = (H + 2 * pad - k) // st + 1
    gen = torch.Generator(device=dev).manual_seed(5000 + Cin + 7 * Cout + k)
    rnd = lambda *shape: torch.randn(*shape, generator=gen, device=dev)                  # noqa: E731
    x16 = rnd(N5, H, H, Cin).to(BF)
    w = rnd(Cout, k, k, Cin) / (Cin * k * k) ** 0.5
    dy16 = rnd(N5, Ho, Ho, Cout).to(BF)
    add16 = rnd(N5, H, H, Cin).to(BF)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N5 * H * H * Cin // 32,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    prev = K.set_conv_arith('bf16')
    try:
        g, g32 = (K.make_geom(N5, H, H, Cin, Cout, k, k, st, pad, T, fold) for _ in range(2))
        y16 = K.conv_fprop(x16, w, g)
        y32 = K.conv_fprop(x16.float(), w, g32)
        assert y16.dtype == BF and y32.dtype == torch.float32
        _bound_ok(y16, y32, 0.98)
        dx16 = K.conv_dgrad(dy16, w, g, add_src=add16, add_mask_src=mask)
        dx32 = K.conv_dgrad(dy16.float(), w, g32, add_src=add16.float(), add_mask_src=mask)
        _bound_ok(dx16, dx32, 0.98)
        dw16 = K.conv_wgrad(dy16, x16, g)
        dw32 = K.conv_wgrad(dy16.float(), x16.float(), g32)
        assert dw16.dtype == torch.float32
        assert (dw16 - dw32).abs().max().item() <= 2e-5 * (dw32.abs().max().item() + 1e-12)
        if site in CPU_CHECKED:
            xc = x16.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
            wc = w.to(BF).float().cpu().permute(0, 3, 1, 2).contiguous()              # the kernels round the weight to bf16 too
            yc = F.conv2d(temporal_shift(xc, 8, 8) if shift else xc, wc, stride=st, padding=pad)
            yc.backward(dy16.float().cpu().permute(0, 3, 1, 2))
            _bound_ok(y16.cpu().permute(0, 3, 1, 2), yc.detach())
            bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder='little').astype(bool).reshape(N5, H, H, Cin)
            dxc = xc.grad.permute(0, 2, 3, 1) + add16.float().cpu() * torch.from_numpy(bits)
            _bound_ok(dx16.cpu(), dxc)
    finally:
        K.set_conv_arith('bf16x3')
        K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


# ---------------------------------------------------------------------------------------------------------------------
# config 4: I3D-R50 at 16 clips x 32 x 224 x 224
# ---------------------------------------------------------------------------------------------------------------------
B4 = 16
# (frames per clip, H = W, Cin, Cout) of the inflated 3x1x1 conv1 sites (inflate=((1,1,1),(1,0,1,0),(1,0,1,0,1,0),(0,1,0)))
I3D_TEMPORAL = [(8, 56, 64, 64), (8, 56, 256, 64), (4, 56, 256, 128), (4, 28, 512, 128), (4, 28, 512, 256), (4, 14, 1024, 256),
                (4, 7, 2048, 512)]
# (frames N, (Cin, Cout, k, stride, H, shift = 0)) of the per-frame convolutions: layer1 runs on 16 x 8 frames, the rest on 16 x 4
I3D_SPATIAL = [(128, (64, 64, 3, 1, 56, 0)), (128, (64, 256, 1, 1, 56, 0)),
               (64, (128, 128, 3, 2, 56, 0)), (64, (128, 512, 1, 1, 28, 0)), (64, (256, 512, 1, 2, 56, 0)), (64, (512, 128, 1, 1, 28, 0)),
               (64, (128, 128, 3, 1, 28, 0)),
               (64, (256, 256, 3, 2, 28, 0)), (64, (256, 1024, 1, 1, 14, 0)), (64, (512, 1024, 1, 2, 28, 0)), (64, (1024, 256, 1, 1, 14, 0)),
               (64, (256, 256, 3, 1, 14, 0)),
               (64, (1024, 512, 1, 1, 14, 0)), (64, (512, 512, 3, 2, 14, 0)), (64, (512, 2048, 1, 1, 7, 0)), (64, (1024, 2048, 1, 2, 14, 0)),
               (64, (2048, 512, 1, 1, 7, 0)), (64, (512, 512, 3, 1, 7, 0))]


@pytest.mark.parametrize('n,site', I3D_SPATIAL, ids=lambda v: 'x'.join(map(str, v)) if isinstance(v, tuple) else f'N{v}')
def test_config4_i3d_spatial_sites(n, site, dev):
    check_site(site, n, dev)


@pytest.mark.parametrize('shape', I3D_TEMPORAL, ids=lambda s: 'x'.join(map(str, s)))
def test_config4_i3d_temporal_sites(shape, dev):
    from bdvcil_amd import kernels as K
    T, H, Cin, Cout = shape
    B, W = B4, H
    gen = torch.Generator().manual_seed(4000 + Cin + 3 * Cout + H)
    x = torch.randn(B, Cin, T, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 1, 1, generator=gen) / (3 * Cin) ** 0.5).requires_grad_(True)
    y = F.conv3d(x, w, padding=(1, 0, 0))
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    frames = lambda t: t.detach().permute(0, 2, 3, 4, 1).reshape(B * T, H, W, -1).contiguous()      # noqa: E731
    xd, dyd = frames(x).to(dev), frames(dy).to(dev)
    wd = w.detach().permute(0, 2, 3, 4, 1).reshape(Cout, 3, 1, Cin).contiguous().to(dev)
    add = torch.randn(B * T, H, W, Cin, generator=gen)
    m = torch.randn(B * T, H, W, Cin, generator=gen) > 0
    bits = torch.from_numpy(np.packbits(m.numpy().reshape(-1), bitorder='little').view(np.int32).copy()).to(dev)
    dx_ref = frames(x.grad) + add * m
    dw_ref = w.grad.permute(0, 2, 3, 4, 1).reshape(Cout, 3, 1, Cin)
    # fp64 dot products for sampled weight-gradient entries
    sg = torch.Generator().manual_seed(6)
    co, ci, dt = (torch.randint(0, hi, (256,), generator=sg) for hi in (Cout, Cin, 3))
    xp = F.pad(x.detach(), (0, 0, 0, 0, 1, 1)).double()
    dy64 = dy.double()
    smp = torch.stack([(xp[:, ci[q], dt[q]:dt[q] + T] * dy64[:, co[q]]).sum() for q in range(256)])
    for mode in ('bf16x3', 'f32mfma'):
        prev = K.set_conv_arith(mode)
        try:
            g = K.make_temporal_geom(B, T, H, W, Cin, Cout, 3)
            e = _err(K.conv_fprop(xd, wd, g).view(B * T, H, W, Cout).cpu(), frames(y))
            assert e <= 2e-5, (mode, 'fprop', e)
            e = _err(K.conv_dgrad(dyd, wd, g, add_src=add.to(dev), add_mask_src=bits).view(B * T, H, W, Cin).cpu(), dx_ref)
            assert e <= 2e-5, (mode, 'dgrad', e)
            dwo = K.conv_wgrad(dyd, xd, g).cpu()
            e = _err(dwo, dw_ref)
            assert e <= 1e-4, (mode, 'wgrad vs fp32 CPU', e)
            es = (dwo[co, dt, 0, ci].double() - smp).abs().max().item() / dw_ref.abs().max().item()
            assert es <= 2e-5, (mode, 'wgrad vs fp64 samples', es)
        finally:
            K.set_conv_arith('bf16x3')
            K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


def test_config4_i3d_stem_full_size(dev):
    """5x7x7 / (2,2,2) stem on 16 x 32 x 224 x 224: ONE launch with temporal taps, forward and weight gradient (the stem needs no
    input gradient) against torch's CPU Conv3d."""
    from bdvcil_amd import kernels as K
    B, T, H, kt, Cout = B4, 32, 224, 5, 64
    To = T // 2
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, 3, T, H, H, generator=gen)
    w = (torch.randn(Cout, 3, kt, 7, 7, generator=gen) / (3 * kt * 49) ** 0.5).requires_grad_(True)
    y = F.conv3d(x, w, stride=(2, 2, 2), padding=(kt // 2, 3, 3))
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    x4 = torch.zeros(B * T, H, H, 4)
    x4[..., :3] = x.permute(0, 2, 3, 4, 1).reshape(B * T, H, H, 3)
    w4 = torch.zeros(Cout, kt, 7, 7, 4)
    w4[..., :3] = w.detach().permute(0, 2, 3, 4, 1)
    g = K.make_geom(B * To, H, H, 4, Cout, 7, 7, 2, 3, T=To, rt=kt, st_t=2)
    x4d = x4.to(dev)
    yd = K.conv_fprop(x4d, w4.view(Cout, kt * 7, 7, 4).to(dev), g)
    ref = y.detach().permute(0, 2, 3, 4, 1).reshape(B * To, g.Ho, g.Wo, Cout)
    e = _err(yd.cpu(), ref)
    assert e <= 2e-5, ('fprop', e)
    dyd = dy.permute(0, 2, 3, 4, 1).reshape(B * To, g.Ho, g.Wo, Cout).contiguous().to(dev)
    dw = K.conv_wgrad(dyd, x4d, g).cpu().view(Cout, kt, 7, 7, 4)
    dref = w.grad.permute(0, 2, 3, 4, 1)                                    # (Cout, kt, 7, 7, 3)
    assert dw[..., 3].abs().max().item() == 0
    e = _err(dw[..., :3], dref)
    assert e <= 1e-4, ('wgrad vs fp32 CPU', e)                               # a reduction over 3.2 M output pixels
    # 256 sampled entries against fp64 dot products
    sg = torch.Generator().manual_seed(7)
    co, c3, dt, rr, ss = (torch.randint(0, hi, (256,), generator=sg) for hi in (Cout, 3, kt, 7, 7))
    xp = F.pad(x, (3, 3, 3, 3, kt // 2, kt // 2)).double()
    dy64 = dy.double()
    Ho = g.Ho
    smp = torch.stack([(xp[:, c3[q], dt[q]:dt[q] + 2 * (To - 1) + 1:2, rr[q]:rr[q] + 2 * (Ho - 1) + 1:2, ss[q]:ss[q] + 2 * (Ho - 1) + 1:2]
                        * dy64[:, co[q]]).sum() for q in range(256)])
    es = (dw[co, dt, rr, ss, c3].double() - smp).abs().max().item() / dref.abs().max().item()
    assert es <= 2e-5, ('wgrad vs fp64 samples', es)


# ---------------------------------------------------------------------------------------------------------------------
# config 3: one CIL task-1 step at B = 32
# ---------------------------------------------------------------------------------------------------------------------
def test_config3_cil_step_full_size_properties(dev):
    import bdvcil_amd as bd
    from bench import model_cfg
    B, K_ = 32, 101
    names = ['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool']
    weights, scale = [0.01] * 5, [1.0, 3.3466401061363023]              # configs/ucf101/bgmix_plus_randAug/...:88-89
    torch.manual_seed(0)
    model = bd.build_model(model_cfg(50, K_, 'LocalSimilarityClassifier', 'LSCLoss', 0.0)).to(dev)
    torch.manual_seed(1)
    prev = bd.build_model(model_cfg(50, K_, 'LocalSimilarityClassifier', 'LSCLoss', 0.0)).to(dev)
    prev.eval()
    for q in prev.parameters():
        q.requires_grad_(False)
    g = torch.Generator().manual_seed(1000)
    labels = torch.randint(0, K_, (B, 1), generator=g).to(dev)
    frames = torch.randint(0, 256, (B, 8, 224, 224, 3), generator=g, dtype=torch.uint8).to(dev)
    bg = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).to(dev)
    mix = (torch.rand(B, generator=torch.Generator().manual_seed(1)) < 0.25).to(dev)
    cur_hooks, prev_hooks = bd.OutputHook(model, names), bd.OutputHook(prev, names)
    front = bd.BackgroundMixFrontEnd(alpha=0.5)
    model.train()

    def step(loss_scale):
        for p in model.parameters():
            p.grad = None
        bn = {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}
        data = dict(imgs=front(frames, bg, mix), label=labels)
        out = bd.base_training_step(model, data, current_task=1, prev_model=prev, current_hooks=cur_hooks, prev_hooks=prev_hooks,
                                    kd_modules_names=names, kd_weight_by_module=weights, adaptive_scale_factors=scale)
        # the KD terms against an independent torch reduction over the hooked tensors (fp64 accumulation, on the device)
        direct = {}
        for n in names:
            c, q = cur_hooks.get_layer_output(n).detach(), prev_hooks.get_layer_output(n).detach()
            assert c.shape == q.shape and c.shape[0] == B * 8
            direct[n] = float(((c.double() - q.double()) ** 2).mean())
        (out['loss'] * loss_scale).backward()
        model.load_state_dict(bn, strict=False)                     # same running statistics for the next call
        vals = {k: float(v.detach()) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
        return vals, direct, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    v1, d1, g1 = step(1.0)
    v1b, _, g1b = step(1.0)
    v2, _, g2 = step(2.0)
    assert v1 == v1b == v2 and all(np.isfinite(x) for x in v1.values())
    for n in names:
        assert d1[n] > 0 and abs(v1[n] - d1[n]) <= 2e-5 * d1[n], (n, v1[n], d1[n])
    kd = scale[1] * sum(w * v1[n] for w, n in zip(weights, names))
    assert abs(v1['kd_loss'] - kd) <= 1e-5 * abs(kd)
    assert abs(v1['loss'] - (v1['loss_cls'] + v1['kd_loss'])) <= 1e-6 * abs(v1['loss'])
    assert set(g1) == set(g1b) == set(g2) and len(g1) > 160
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
        assert torch.equal(g1[n], g1b[n]), n                        # run-to-run deterministic (no atomics on the path)
        assert torch.equal(g2[n], g1[n] * 2), n                     # backward is linear in the loss gradient
    assert float(g1['cls_head.loss_cls.eta'].abs()) > 0
    # the teacher's eval forward (running statistics): clips are independent of their batch
    with torch.no_grad():
        imgs = front(frames, bg, mix)
        full = prev(imgs, return_loss=False)
        part = prev(front(frames[:4].contiguous(), bg[:4].contiguous(), mix[:4].contiguous()), return_loss=False)
    assert full.shape == (B, K_)
    assert (full[:4] - part).abs().max().item() <= 1e-4 * max(1.0, full.abs().max().item())
    assert torch.equal(full[:4].argmax(1), part.argmax(1))
    cur_hooks.remove()
    prev_hooks.remove()
