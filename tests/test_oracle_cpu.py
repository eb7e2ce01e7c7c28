"""CPU tests of the oracle itself: pinned against the golden vectors generated from the reference's own
cosine_linear.py / inc_net.py / lsc_loss.py (tests/golden/make_golden.py), plus definition-level checks of the
UPSTREAM restatements (temporal shift by slicing, param-group census of SURVEY section 8(c)(6))."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tsm_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'head_loss_golden.npz')


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_lsc_and_lscloss_match_reference_golden():
    gz = np.load(GOLD)
    assert int(gz['n_lsc']) == 5
    for i in range(int(gz['n_lsc'])):
        p = f'lsc{i}_'
        x = _t(gz[p + 'x']).requires_grad_(True)
        w = _t(gz[p + 'w']).requires_grad_(True)
        eta = _t(gz[p + 'eta']).requires_grad_(True)
        y = _t(gz[p + 'y'])
        K, P = w.shape[0], int(gz[p + 'P'])
        sim = O.lsc_forward(x, w, K, P)
        sim.retain_grad()
        loss = O.lsc_loss(sim, y, eta)
        loss.backward()
        assert torch.allclose(sim, _t(gz[p + 'sim']), rtol=0, atol=1e-6)
        assert torch.allclose(loss, _t(gz[p + 'loss']), rtol=1e-6, atol=1e-7)
        assert torch.allclose(sim.grad, _t(gz[p + 'dsim']), rtol=1e-5, atol=1e-8)
        assert torch.allclose(x.grad, _t(gz[p + 'dx']), rtol=1e-4, atol=1e-8)
        assert torch.allclose(w.grad, _t(gz[p + 'dw']), rtol=1e-4, atol=1e-8)
        assert torch.allclose(eta.grad, _t(gz[p + 'deta']), rtol=1e-5, atol=1e-7)
        assert bool(gz[p + 'grown_old_rows_kept']) and tuple(gz[p + 'grown_shape']) == (K + 5, w.shape[1])


def test_lscloss_hinge_corner_matches_reference_golden():
    gz = np.load(GOLD)
    sim = _t(gz['hinge_sim']).requires_grad_(True)
    eta = torch.tensor([10.0], requires_grad=True)
    loss = O.lsc_loss(sim, _t(gz['hinge_y']), eta)
    loss.backward()
    assert torch.allclose(loss, _t(gz['hinge_loss']), rtol=1e-6)
    assert torch.allclose(sim.grad, _t(gz['hinge_dsim']), rtol=1e-5, atol=1e-8)
    assert torch.allclose(eta.grad, _t(gz['hinge_deta']), rtol=1e-5, atol=1e-8)


def test_incremental_net_matches_reference_golden():
    gz = np.load(GOLD)
    for i in range(int(gz['n_inc'])):
        p = f'inc{i}_'
        net = O.IncrementalNet(gz[p + 'x'].shape[1], gz[p + 'w'].shape[0])
        with torch.no_grad():
            net.weight.copy_(_t(gz[p + 'w']))
            net.bias.copy_(_t(gz[p + 'b']))
        x = _t(gz[p + 'x']).requires_grad_(True)
        out = net(x)
        out.backward(_t(gz[p + 'dy']))
        assert torch.allclose(out, _t(gz[p + 'out']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(x.grad, _t(gz[p + 'dx']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(net.weight.grad, _t(gz[p + 'dw']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(net.bias.grad, _t(gz[p + 'db']), rtol=1e-5, atol=1e-6)
        K = net.out_features
        net.update_fc(K + 3)
        assert torch.equal(net.weight.detach()[:K], _t(gz[p + 'w'])) and bool(gz[p + 'grown_old_rows_kept'])
        assert torch.equal(net.bias.detach(), _t(gz[p + 'grown_b']))


def test_temporal_shift_by_slicing_definition():
    """SURVEY section 8(c) golden (1): x (2*8, 16, 3, 3) arange, expected by the slicing definition."""
    T, C = 8, 16
    x = torch.arange(2 * T * C * 9, dtype=torch.float32).view(2 * T, C, 3, 3)
    out = O.temporal_shift(x, T, 8)
    v, o = x.view(2, T, C, 9), out.view(2, T, C, 9)
    fold = C // 8
    assert torch.equal(o[:, :-1, :fold], v[:, 1:, :fold]) and o[:, -1, :fold].abs().sum() == 0
    assert torch.equal(o[:, 1:, fold:2 * fold], v[:, :-1, fold:2 * fold]) and o[:, 0, fold:2 * fold].abs().sum() == 0
    assert torch.equal(o[:, :, 2 * fold:], v[:, :, 2 * fold:])
    # the shift never crosses a clip boundary
    x2 = x.clone()
    x2[T:] += 1000.0
    assert torch.equal(O.temporal_shift(x2, T, 8)[:T], out[:T])


@pytest.mark.parametrize('depth,n_conv,n_bn_tensors,n_blocks', [(18, 20, 40, 8), (34, 36, 72, 16), (50, 53, 106, 16)])
def test_param_group_census(depth, n_conv, n_bn_tensors, n_blocks):
    """SURVEY section 8(c) golden (6) / Appendix B: conv counts, BN tensors, shift sites, group hyper-parameters."""
    m = O.build_model(O.r50_cfg(num_classes=7, depth=depth, head='LocalSimilarityClassifier', loss='LSCLoss'))
    groups = O.param_groups(m, 0.01, 1e-4, 5.0)
    assert [len(g['params']) for g in groups] == [1, 0, n_conv - 1, 0, n_bn_tensors, 2, 0]
    assert [g['lr'] for g in groups] == [0.01, 0.02, 0.01, 0.02, 0.01, 0.05, 0.1]
    assert [g['weight_decay'] for g in groups] == [1e-4, 0, 1e-4, 0, 0, 1e-4, 0]
    assert sum(isinstance(x, O.TemporalShift) for x in m.modules()) == n_blocks
    keys = list(m.state_dict().keys())
    assert 'backbone.conv1.conv.weight' in keys and 'backbone.layer1.0.conv1.conv.net.weight' in keys
    assert 'cls_head.fc_cls.weights' in keys and 'cls_head.loss_cls.eta' in keys
    m2 = O.build_model(O.r50_cfg(num_classes=7, depth=depth, head='SimpleLinear', loss='CrossEntropyLoss'))
    assert [len(g['params']) for g in O.param_groups(m2, 0.01, 1e-4, 5.0)] == [1, 0, n_conv - 1, 0, n_bn_tensors, 1, 1]


def test_r50_parameter_count_matches_metafile():
    """configs/recognition/tsm/metafile.yml:15 (24 327 632 params at K=400) = 23 508 032 backbone + 2048*400+400."""
    m = O.build_model(O.r50_cfg(num_classes=400, depth=50, head='SimpleLinear', loss='CrossEntropyLoss'))
    n_backbone = sum(p.numel() for p in m.backbone.parameters())
    n_head = sum(p.numel() for p in m.cls_head.fc_cls.parameters())
    assert n_backbone == 23508032 and n_backbone + n_head == 24327632


def test_r50_flop_count_matches_metafile():
    """configs/recognition/tsm/metafile.yml:14 (FLOPs: 32965562368 for tsm_r50_1x1x8 at 8 x 224 x 224, K = 400), the only
    other backbone fact the reference tree holds.  The figure is mmcv's ``get_model_complexity_info`` count: conv MACs +
    2 per BatchNorm element + 1 per ReLU element + max-/avg-pool input elements + classifier MACs.  The oracle's layer
    shapes, its BatchNorm and ReLU placement and the head reproduce it exactly: 32 697 090 048 conv MACs (= 8 x 4.0871 G,
    SURVEY Appendix B, the figure bench.py's algorithmic FLOP are built on) + 177 823 744 + 76 869 632 + 7 225 344 + 6 553 600."""
    import torch.nn as nn
    m = O.build_model(O.r50_cfg(num_classes=400, depth=50, head='SimpleLinear', loss='CrossEntropyLoss'))
    m.eval()
    cnt = dict(conv=0, bn=0, relu=0, pool=0)

    def conv_h(mod, i, o):
        cnt['conv'] += o.numel() * mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]

    def bn_h(mod, i, o):
        cnt['bn'] += 2 * i[0].numel()

    def pool_h(mod, i, o):
        cnt['pool'] += i[0].numel()

    class _F:                                   # the oracle applies ReLU through its module-level ``F``
        def __init__(self, real):
            self._real = real

        def relu(self, x, *a, **kw):
            cnt['relu'] += x.numel() if x.dim() == 4 else 0
            return self._real.relu(x, *a, **kw)

        def __getattr__(self, name):
            return getattr(self._real, name)

    for x in m.modules():
        if isinstance(x, nn.Conv2d):
            x.register_forward_hook(conv_h)
        elif isinstance(x, nn.BatchNorm2d):
            x.register_forward_hook(bn_h)
        elif isinstance(x, nn.MaxPool2d):
            x.register_forward_hook(pool_h)
    real, O.F = O.F, _F(O.F)
    try:
        with torch.no_grad():
            m(torch.zeros(1, 8, 3, 224, 224), return_loss=False)
    finally:
        O.F = real
    avgpool, fc = 8 * 2048 * 7 * 7, 8 * 2048 * 400
    assert cnt['conv'] == 32697090048
    assert cnt['conv'] + cnt['bn'] + cnt['relu'] + cnt['pool'] + avgpool + fc == 32965562368


def test_bgmix_formula():
    g = torch.Generator().manual_seed(0)
    fr = torch.randint(0, 256, (2, 3, 5, 6, 3), generator=g, dtype=torch.uint8)
    bg = torch.randint(0, 256, (2, 5, 6, 3), generator=g, dtype=torch.uint8)
    out = O.bgmix_normalize(fr, bg, torch.tensor([True, False]), 0.5)
    mean, std = torch.tensor(O.IMG_MEAN), torch.tensor(O.IMG_STD)
    x = (fr.float() - mean) / std
    b = (bg.float() - mean) / std
    exp0 = (0.5 * x[0] + 0.5 * b[0][None]).permute(0, 3, 1, 2)
    assert out.shape == (2, 3, 3, 5, 6)
    assert torch.allclose(out[0], exp0, rtol=1e-5, atol=1e-5)
    assert torch.allclose(out[1], x[1].permute(0, 3, 1, 2), rtol=1e-5, atol=1e-5)


def test_forward_shapes_and_b1_squeeze():
    """Appendix C.3: labels.squeeze() turns a B=1 batch into a 0-d tensor; BaseHead.loss re-unsqueezes."""
    m = O.build_model(O.r50_cfg(num_classes=5, depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0))
    imgs = torch.randn(1, 8, 3, 32, 32)
    out = m(imgs, torch.tensor([[3]]))
    assert set(out) == {'top1_acc', 'top5_acc', 'loss_cls'} and out['loss_cls'].dim() == 0
    m.eval()
    with torch.no_grad():
        p = m(imgs, return_loss=False)
    assert p.shape == (1, 5) and abs(p.sum().item() - 1.0) < 1e-5          # average_clips='prob'


def test_acm_smooth_ce_matches_reference_golden():
    gz = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'acm_golden.npz'))
    for i in range(int(gz['n'])):
        p = f'c{i}_'
        score = _t(gz[p + 'score']).requires_grad_(True)
        K = score.shape[1]
        loss = O.acm_smooth_ce(score, _t(gz[p + 'labels']), _t(gz[p + 'bg']), _t(gz[p + 'fg']), K, float(gz[p + 'alpha']))
        loss.backward()
        assert torch.allclose(loss, _t(gz[p + 'loss']), rtol=1e-6, atol=1e-7)
        assert torch.allclose(score.grad, _t(gz[p + 'dscore']), rtol=1e-5, atol=1e-8)


def test_i3d_oracle_and_plugin_surface():
    """oracle/i3d_oracle.py (parity unpinned) against the facts available here: 33.13 GMAC of convolutions per 32 x 224 x 224
    clip ("about 33 GMAC/clip by my count"), the inflation pattern of configs/_base_/models/i3d_r50.py:15
    (3 + 2 + 3 + 1 inflated blocks), and a state_dict that the product's Recognizer3D -- built from that config through the
    registry -- loads key for key."""
    import torch.nn as nn
    from oracle import i3d_oracle as I
    import bdvcil_amd as bd
    assert I.i3d_conv_macs(32, 224) == 33127399424
    ref = I.Recognizer3D(400)
    inflated = [isinstance(m, nn.Conv3d) and m.kernel_size == (3, 1, 1) for m in ref.modules()]
    assert sum(inflated) == 9
    cfg = dict(type='Recognizer3D',
               backbone=dict(type='ResNet3d', pretrained2d=True, pretrained=None, depth=50, conv1_kernel=(5, 7, 7), conv1_stride_t=2,
                             pool1_stride_t=2, conv_cfg=dict(type='Conv3d'), norm_eval=False,
                             inflate=((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0)), zero_init_residual=False),
               cls_head=dict(type='I3DHead', num_classes=400, in_channels=2048, spatial_type='avg', dropout_ratio=0.5, init_std=0.01),
               train_cfg=None, test_cfg=dict(average_clips='prob'))
    mod = bd.build_model(cfg)
    assert set(mod.state_dict()) == set(ref.state_dict())
    mod.load_state_dict(ref.state_dict())
    assert sum(p.numel() for p in mod.parameters()) == sum(p.numel() for p in ref.parameters()) == 27223872 + 2048 * 400 + 400
    with pytest.raises(FileNotFoundError):
        bd.build_model(dict(cfg, backbone=dict(cfg['backbone'], pretrained='torchvision://resnet50')))


def test_bg_resize_crop_oracle_sizes_and_antialias():
    """``O.bg_resize_crop`` (Resize(256) -> RandomCrop(224) of libs/loader/comix_loader.py:72-73): torchvision's output-size rule,
    the crop window, and the fact the build relies on -- enlarging an image, the antialiased and the plain bilinear forms coincide."""
    g = torch.Generator().manual_seed(0)
    for (h, w) in [(240, 320), (320, 240), (256, 256)]:
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        a = O.bg_resize_crop(img, 256, 224, 3, 5)
        b = O.bg_resize_crop(img, 256, 224, 3, 5, antialias=True)
        assert a.shape == (224, 224, 3) and a.dtype == torch.float32
        assert (a - b).abs().max().item() <= 1e-4
        assert 0.0 <= float(a.min()) and float(a.max()) <= 255.0
    same = torch.randint(0, 256, (256, 300, 3), generator=g, dtype=torch.uint8)        # smaller edge already 256: Resize is the identity
    assert torch.equal(O.bg_resize_crop(same, 256, 224, 0, 10), same[0:224, 10:234].float())
