"""I3D-ResNet50 (BASELINE config 4) on the HIP kernels against the CPU restatement
oracle/i3d_oracle.py (parity unpinned: mmaction2's ResNet3d is not vendored and the reference holds no fixture for it).

Bars: kernels 2e-5 of the output scale (fp32 both sides); model logits within 1e-3 with identical arg-max; training step: loss
1e-4 relative, parameter gradients by relative L2 error against the fp64 oracle, no further from it than 3x the fp32 CPU oracle
plus 1e-4 -- except upstream of a COUNTED flip (a ReLU sign, a pool1 arg-max or a pool2 winner that differs from the fp64 oracle's
where fp32 rounding can reach it; read back and counted as in test_model_gpu.py), where the floor is 3e-2: at this test's clip
size the BatchNorm populations of layer3 / layer4 are tens of values per channel, so one flipped sign moves a channel's gradient
by per cents."""
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


class _ReluRecorder3d:
    """Arguments of every F.relu call of the I3D oracle's backbone (5-D tensors), in execution order: stem, then per block conv1,
    conv2, block output.  The oracle calls ``F.relu`` through its module-level ``F`` (oracle/i3d_oracle.py)."""

    class _Proxy:
        def __init__(self, real, sink):
            self._real, self._sink = real, sink

        def relu(self, x, *a, **kw):
            if x.dim() == 5:
                self._sink.append(x.detach())
            return self._real.relu(x, *a, **kw)

        def __getattr__(self, name):
            return getattr(self._real, name)

    def __enter__(self):
        self.pre, self._saved = [], I.F
        I.F = _ReluRecorder3d._Proxy(self._saved, self.pre)
        return self

    def __exit__(self, *exc):
        I.F = self._saved
        return False


def _frames(t):
    """(B, C, T, H, W) -> (B*T, C, H, W): the frame-major order of the HIP path's NHWC storage"""
    B, C, T, H, W = t.shape
    return t.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)


def _i3d_site_owners(ref):
    sites = [['backbone.conv1.']]
    for li in range(1, 5):
        for bi, _ in enumerate(getattr(ref.backbone, f'layer{li}')):
            pre = f'backbone.layer{li}.{bi}.'
            sites += [[pre + 'conv1.'], [pre + 'conv2.'], [pre + 'conv3.', pre + 'downsample.']]
    return sites


def _count_pool2_flips(pre64, pre32, sel):
    """pool2 = MaxPool3d((2,1,1)): which frame of a pair wins is as discontinuous as a ReLU sign.  ``sel`` (bdv_maxpool_t2_fwd): one
    bit per output element, set when the second frame won.  Differences from the fp64 oracle are legitimate only where the two
    candidates are within max(1e-5, 3 x the fp32 CPU oracle's deviation) of the channel scale of each other; pairs whose maximum
    is <= 0 are left aside (both candidates are zero after the ReLU: a tie, and the ReLU masks the gradient wherever it lands)."""
    import numpy as np
    act = pre64.clamp_min(0)                                     # (B, C, T, H, W): layer1's output
    B, C, T, H, W = act.shape
    a, b = act[:, :, 0::2], act[:, :, 1::2]
    want = (b > a).permute(0, 2, 3, 4, 1).reshape(-1)           # (B, T/2, H, W, C) flat = the HIP output order
    bits = torch.from_numpy(np.unpackbits(sel.cpu().numpy().view(np.uint8), bitorder='little').astype(bool))
    live = (torch.maximum(a, b) > 0).permute(0, 2, 3, 4, 1).reshape(-1)
    diff = (bits != want) & live
    n = int(diff.sum())
    if n:
        cmax = pre64.abs().amax(dim=(0, 2, 3, 4), keepdim=True)
        noise32 = float(((pre32.double() - pre64).abs() / cmax).max())
        gap = ((a - b).abs() / cmax).permute(0, 2, 3, 4, 1).reshape(-1)[diff]
        bar = max(1e-5, 3 * noise32)
        assert float(gap.max()) <= bar, f'pool2: the winner differs where the candidates are {float(gap.max()):.2e} of the channel scale apart'
    return n


@pytest.mark.parametrize('T,S,B', [(8, 64, 2), (16, 96, 2)])
def test_i3d_train_step(T, S, B, dev, conv_arith):
    """Loss, accuracies, every parameter gradient and the BatchNorm statistics of one training step, judged as
    tests/test_model_gpu.py::test_train_step judges the TSM path: the ReLU sign bits the HIP path wrote, pool1's arg-max codes and
    pool2's winners are read back and compared with the fp64 oracle's; a difference must sit where fp32 rounding can reach it, and
      * parameters with NO counted flip behind them (in backward order) are held to  relL2 <= 3 * e_f32 + 1e-4,
      * parameters upstream of a counted flip keep the looser  3 * e_f32 + 3e-2  (at this clip size the BatchNorm populations of
        layer3 / layer4 are tens of values per channel: one flipped sign moves a channel's gradient by per cents)."""
    import bdvcil_amd as bd
    from bdvcil_amd import functional as Fn
    from test_model_gpu import _report, count_pool_flips, count_relu_flips     # (tests/ is on sys.path: pytest rootdir import)
    K_ = 9
    ref, mod = _pair(K_, dev, seed=2)
    ref64 = copy.deepcopy(ref).double()
    gen = torch.Generator().manual_seed(7)
    imgs = torch.randn(B, 1, 3, T, S, S, generator=gen)
    labels = torch.randint(0, K_, (B, 1), generator=gen)
    ref.train(); mod.train(); ref64.train()
    with _ReluRecorder3d() as rec32:
        rl = ref(imgs, labels)
    rl['loss_cls'].backward()
    with _ReluRecorder3d() as rec:
        r64 = ref64(imgs.double(), labels)
    r64['loss_cls'].backward()
    Fn.RELU_MASK_TAP = taps = []
    Fn.POOL_IDX_TAP = pools = []
    try:
        ol = mod(imgs.to(dev), labels.to(dev))
    finally:
        Fn.RELU_MASK_TAP = Fn.POOL_IDX_TAP = None
    ol['loss_cls'].backward()
    # layer4's BatchNorms normalise over few samples here: the fp32 CPU oracle itself is 3.5e-4 away from its fp64 run in the loss
    # and 1.1e-4 (relative) in a running variance.  Every bar below is therefore stated against the fp64 oracle: 3x the fp32 CPU
    # oracle's own distance from it, plus the plain tolerance.
    l64 = r64['loss_cls'].item()
    assert abs(ol['loss_cls'].item() - l64) <= 3 * abs(rl['loss_cls'].item() - l64) + 1e-4 * max(1.0, abs(l64))
    assert abs(ol['top1_acc'].item() - rl['top1_acc'].item()) < 1e-6
    pre64, pre32 = [_frames(p) for p in rec.pre], [_frames(p) for p in rec32.pre]
    flips, flips32 = count_relu_flips(pre64, pre32, taps)
    owners = _i3d_site_owners(ref)
    assert len(owners) == len(flips) == 1 + 3 * 16
    last_flip = max([k for k, n in enumerate(flips) if n], default=-1)
    assert len(pools) == 2                                  # pool1's arg-max codes, pool2's winners
    # pool1 = the spatial 3x3 / 2 max-pool of the even frames of the stem activation
    Bc, C0, To, H0, W0 = rec.pre[0].shape
    even = lambda t: t[:, :, 0::2].permute(0, 2, 1, 3, 4).reshape(-1, C0, H0, W0)      # noqa: E731
    pool1_flips = count_pool_flips(even(rec.pre[0]), even(rec32.pre[0]), pools[0])
    pool2_flips = _count_pool2_flips(rec.pre[9], rec32.pre[9], pools[1])                 # site 9 = layer1's output
    if pool1_flips:
        last_flip = max(last_flip, 0)
    if pool2_flips:
        last_flip = max(last_flip, 9)
    _report(f'[relu flips] I3D T={T} S={S} B={B} {conv_arith}: {sum(flips)} of {sum(p.numel() for p in rec.pre)} signs differ from the '
            f'fp64 oracle at sites {[k for k, n in enumerate(flips) if n]} (fp32 CPU oracle: {sum(flips32)} at '
            f'{[k for k, n in enumerate(flips32) if n]}); pool1 arg-max differs at {pool1_flips}, pool2 winner at {pool2_flips} elements')

    def behind_a_flip(name):
        for k, prefixes in enumerate(owners):
            if any(name.startswith(q) for q in prefixes):
                return k <= last_flip
        return False                                        # head: after every ReLU of the backbone
    rp, r64p, op = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(mod.named_parameters())
    worst = {}
    for name, p in rp.items():
        assert op[name].grad is not None, name
        assert tuple(op[name].grad.shape) == tuple(p.shape), name
        e_hip, e_f32 = _rel_l2(op[name].grad, r64p[name].grad), _rel_l2(p.grad, r64p[name].grad)
        loose = behind_a_flip(name)
        assert e_hip <= 3 * e_f32 + (3e-2 if loose else 1e-4), (name, e_hip, e_f32, 'behind a counted flip' if loose else 'no flip behind it', flips)
        if e_hip > worst.get(loose, (0,))[0]:
            worst[loose] = (e_hip, e_f32, name)
    _report(f'[grad parity] I3D T={T} S={S} B={B} {conv_arith}: worst (relL2 hip, relL2 fp32 CPU, name) with no flip behind: '
            f'{worst.get(False)}; behind a flip: {worst.get(True)}')
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
