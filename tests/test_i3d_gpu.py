"""I3D-ResNet50 (BASELINE config 4) on the HIP kernels against the CPU restatement
oracle/i3d_oracle.py (parity unpinned: mmaction2's ResNet3d is not vendored and the reference holds no fixture for it).

Bars: kernels 2e-5 of the output scale (fp32 both sides); model logits within 1e-3 with identical arg-max; training step: loss
1e-4 relative, parameter gradients by relative L2 error against the fp64 oracle, no further from it than 3x the fp32 CPU oracle
plus 3e-2 (ReLU / max-pool ties flip in any two fp32 implementations, see test_model_gpu.py; at this test's clip size the
BatchNorm populations of layer3 / layer4 are 64 and 16 values per channel, so one flipped sign moves a channel's gradient by
per cents -- observed 1.6e-2 on one layer3 tensor where the fp32 CPU oracle happened to have no flip)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import i3d_oracle as I

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-7, f'max err {err} vs scale {scale}'


@pytest.mark.parametrize('shape', [(2, 8, 7, 7, 64, 64), (3, 4, 6, 5, 256, 128), (2, 8, 14, 14, 128, 256)])
def test_temporal_conv_as_kx1(shape, dev, conv_arith):
    """A 3x1x1 Conv3d (padding (1,0,0)) == the 3x1 convolution on the [B][T][H*W][C] view: fprop, dgrad (with the residual add
    and ReLU mask of a block's conv1) and wgrad against torch.conv3d + autograd."""
    import numpy as np
    from bdvcil_amd import kernels as K
    B, T, H, W, Cin, Cout = shape
    gen = torch.Generator().manual_seed(B * 100 + Cin)
    x = torch.randn(B, Cin, T, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 1, 1, generator=gen) / (3 * Cin) ** 0.5).requires_grad_(True)
    y = F.conv3d(x, w, padding=(1, 0, 0))
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    g = K.make_temporal_geom(B, T, H, W, Cin, Cout, 3)
    frames = lambda t: t.detach().permute(0, 2, 3, 4, 1).reshape(B * T, H, W, -1).contiguous()      # noqa: E731
    xd, dyd = frames(x).to(dev), frames(dy).to(dev)
    wd = w.detach().permute(0, 2, 3, 4, 1).reshape(Cout, 3, 1, Cin).contiguous().to(dev)
    yo = K.conv_fprop(xd, wd, g)
    assert yo.shape == (B, T, H * W, Cout)
    _close(yo.view(B * T, H, W, Cout).cpu(), frames(y))
    add = torch.randn(B * T, H, W, Cin, generator=gen)
    m = torch.randn(B * T, H, W, Cin, generator=gen) > 0
    bits = torch.from_numpy(np.packbits(m.numpy().reshape(-1), bitorder='little').view(np.int32).copy()).to(dev)
    dxo = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev), add_mask_src=bits)
    _close(dxo.view(B * T, H, W, Cin).cpu(), frames(x.grad) + add * m)
    dwo = K.conv_wgrad(dyd, xd, g)
    _close(dwo.cpu(), w.grad.permute(0, 2, 3, 4, 1).reshape(Cout, 3, 1, Cin))
    with pytest.raises(ValueError):
        K.conv_fprop(xd[:, :, :, :Cin // 2].contiguous(), wd, g)          # not a view of the geometry's tensor


def test_maxpool_t2(dev):
    from bdvcil_amd import kernels as K
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 6, 6, 64, generator=gen)                          # (B, T, H, W, C)
    x[0, 0] = x[0, 1]                                                     # ties go to the first frame
    xr = x.permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    ref = F.max_pool3d(xr, (2, 1, 1), (2, 1, 1))
    dout = torch.randn(ref.shape, generator=gen)
    ref.backward(dout)
    out, sel = K.maxpool_t2_fwd(x.view(16, 6, 6, 64).to(dev))
    assert torch.equal(out.cpu().view(2, 4, 6, 6, 64), ref.detach().permute(0, 2, 3, 4, 1))
    dx = K.maxpool_t2_bwd(dout.permute(0, 2, 3, 4, 1).reshape(8, 6, 6, 64).contiguous().to(dev), sel)
    assert torch.equal(dx.cpu().view(2, 8, 6, 6, 64), xr.grad.permute(0, 2, 3, 4, 1))
    with pytest.raises(ValueError):
        K.maxpool_t2_fwd(torch.zeros(3, 6, 6, 64, device=dev))


@pytest.mark.parametrize('B,T,H,W,kt', [(2, 8, 20, 24, 5), (1, 4, 18, 22, 5), (3, 6, 16, 16, 3)])
def test_stem_with_temporal_taps(B, T, H, W, kt, dev):
    """The stem kernels of the bf16-piece arithmetic with temporal taps (bdv_conv_geom.Rt / st_t): ONE launch for the kt x 7 x 7
    filter with strides (2, 2, 2), forward and weight gradient, against torch's CPU Conv3d."""
    from bdvcil_amd import kernels as K
    gen = torch.Generator().manual_seed(B * 100 + T)
    Cout, To = 64, T // 2
    x = torch.randn(B, 3, T, H, W, generator=gen)
    w = (torch.randn(Cout, 3, kt, 7, 7, generator=gen) / (3 * kt * 49) ** 0.5).requires_grad_(True)
    y = F.conv3d(x, w, stride=(2, 2, 2), padding=(kt // 2, 3, 3))
    assert y.shape[2] == To
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    x4 = torch.zeros(B * T, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 4, 1).reshape(B * T, H, W, 3)
    w4 = torch.zeros(Cout, kt, 7, 7, 4)
    w4[..., :3] = w.detach().permute(0, 2, 3, 4, 1)
    g = K.make_geom(B * To, H, W, 4, Cout, 7, 7, 2, 3, T=To, rt=kt, st_t=2)
    yd = K.conv_fprop(x4.to(dev), w4.view(Cout, kt * 7, 7, 4).to(dev), g)
    ref = y.detach().permute(0, 2, 3, 4, 1).reshape(B * To, g.Ho, g.Wo, Cout)
    assert (yd.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    dyd = dy.permute(0, 2, 3, 4, 1).reshape(B * To, g.Ho, g.Wo, Cout).contiguous().to(dev)
    dw = K.conv_wgrad(dyd, x4.to(dev), g).cpu().view(Cout, kt, 7, 7, 4)
    dref = w.grad.permute(0, 2, 3, 4, 1)
    assert (dw[..., :3] - dref).abs().max().item() <= 2e-5 * dref.abs().max().item()
    assert dw[..., 3].abs().max().item() == 0           # the 4th input channel is zero
    with pytest.raises(Exception):                      # only the bf16-piece stem kernels take temporal taps
        K.conv_fprop(x4.to(dev), w4.view(Cout, kt * 7, 7, 4).to(dev), g, x3=False)


def _cfg(K_):
    return dict(type='Recognizer3D',
                backbone=dict(type='ResNet3d', pretrained2d=True, pretrained=None, depth=50, conv1_kernel=(5, 7, 7),
                              conv1_stride_t=2, pool1_stride_t=2, conv_cfg=dict(type='Conv3d'), norm_eval=False,
                              inflate=((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0)), zero_init_residual=False),
                cls_head=dict(type='I3DHead', num_classes=K_, in_channels=2048, spatial_type='avg', dropout_ratio=0.0, init_std=0.01),
                train_cfg=None, test_cfg=dict(average_clips='prob'))         # configs/_base_/models/i3d_r50.py:1-27, dropout 0 for parity


def _pair(K_, dev, seed=0):
    import bdvcil_amd as bd
    torch.manual_seed(seed)
    ref = I.Recognizer3D(K_, dropout_ratio=0.0)
    gen = torch.Generator().manual_seed(seed + 1)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.weight.data.uniform_(0.5, 1.5, generator=gen)
            m.bias.data.normal_(0, 0.1, generator=gen)
            m.running_mean.normal_(0, 0.1, generator=gen)
            m.running_var.uniform_(0.5, 1.5, generator=gen)
    ref.cls_head.fc_cls.weight.data.normal_(0, 0.05, generator=gen)        # init_std 0.01 would leave the logits near zero
    mod = bd.build_model(_cfg(K_))
    mod.load_state_dict(ref.state_dict())
    return ref, mod.to(dev)


def _rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize('T,S,clips', [(8, 64, 2), (16, 96, 1)])
def test_i3d_eval_logits(T, S, clips, dev, conv_arith):
    ref, mod = _pair(9, dev)
    imgs = torch.randn(2, clips, 3, T, S, S, generator=torch.Generator().manual_seed(3))
    ref.eval(); mod.eval()
    with torch.no_grad():
        for mode in ('prob', 'score'):
            ref.test_cfg['average_clips'] = mod.test_cfg['average_clips'] = mode
            r = ref(imgs, return_loss=False)
            o = mod(imgs.to(dev), return_loss=False).cpu()
            assert o.shape == r.shape == (2, 9)
            assert (o - r).abs().max().item() <= 1e-3, (mode, (o - r).abs().max().item())
            assert torch.equal(o.argmax(1), r.argmax(1))


def test_i3d_train_step(dev, conv_arith):
    import bdvcil_amd as bd
    K_ = 9
    ref, mod = _pair(K_, dev, seed=2)
    ref64 = copy.deepcopy(ref).double()
    gen = torch.Generator().manual_seed(7)
    imgs = torch.randn(2, 1, 3, 8, 64, 64, generator=gen)
    labels = torch.randint(0, K_, (2, 1), generator=gen)
    ref.train(); mod.train(); ref64.train()
    rl = ref(imgs, labels)
    rl['loss_cls'].backward()
    r64 = ref64(imgs.double(), labels)
    r64['loss_cls'].backward()
    ol = mod(imgs.to(dev), labels.to(dev))
    ol['loss_cls'].backward()
    # layer4's BatchNorms normalise over 8 samples here (2 clips x 1 frame x 2x2): the fp32 CPU oracle itself is 3.5e-4 away
    # from its fp64 run in the loss and 1.1e-4 (relative) in a running variance.  Every bar below is therefore stated against
    # the fp64 oracle: 3x the fp32 CPU oracle's own distance from it, plus the plain tolerance.
    l64 = r64['loss_cls'].item()
    assert abs(ol['loss_cls'].item() - l64) <= 3 * abs(rl['loss_cls'].item() - l64) + 1e-4 * max(1.0, abs(l64))
    assert abs(ol['top1_acc'].item() - rl['top1_acc'].item()) < 1e-6
    rp, r64p, op = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(mod.named_parameters())
    for name, p in rp.items():
        assert op[name].grad is not None, name
        assert tuple(op[name].grad.shape) == tuple(p.shape), name
        e_hip, e_f32 = _rel_l2(op[name].grad, r64p[name].grad), _rel_l2(p.grad, r64p[name].grad)
        assert e_hip <= 3 * e_f32 + 3e-2, (name, e_hip, e_f32)
    rb, r64b, ob = dict(ref.named_buffers()), dict(ref64.named_buffers()), dict(mod.named_buffers())
    for name, b in rb.items():
        if name.endswith('num_batches_tracked'):
            assert int(ob[name].item()) == int(b.item()), name
        else:
            b64 = r64b[name]
            e_hip = (ob[name].cpu().double() - b64).abs().max().item()
            e_f32 = (b.double() - b64).abs().max().item()
            assert e_hip <= 3 * e_f32 + 1e-4 * b64.abs().max().item() + 1e-6, (name, e_hip, e_f32)
    # one fused SGD step with the reference's parameter groups runs on the 5-D weights
    opt = bd.build_optimizer(mod, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                       paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
    before = mod.backbone.layer3[0].conv1.conv.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, mod.backbone.layer3[0].conv1.conv.weight.detach())
