"""Host-side logic of the product package that needs no GPU: registries, module tree / state_dict layout,
optimizer constructors, hooks, the C-ABI library exports, and the loud failure without a GPU."""
import copy
import ctypes
import os
import pickle
import re

import pytest
import torch
import torch.nn as nn

import bdvcil_amd as bd
from oracle import tsm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', K=7):
    return O.r50_cfg(num_classes=K, depth=depth, head=head, loss=loss)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'bdvcil_hip.h')).read()
    declared = set(re.findall(r'\b(bdv_[a-z0-9_]+)\s*\(', hdr))
    declared.discard('bdv_conv_geom')
    assert len(declared) >= 35
    lib = ctypes.CDLL(bd._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/bdvcil_hip.h but not exported'
    assert declared == set(bd._lib.SIGNATURES), declared ^ set(bd._lib.SIGNATURES)
    assert bd._lib.lib().bdv_abi_version() == bd._lib.ABI_VERSION


def test_conv_geom_mirror_matches_the_header():
    """``_lib.ConvGeom`` (ctypes) lists the fields of ``bdv_conv_geom`` in the header's order, all int32: a field added on one side
    only would shift every later one silently."""
    hdr = open(os.path.join(ROOT, 'include', 'bdvcil_hip.h')).read()
    body = re.search(r'typedef struct bdv_conv_geom \{(.*?)\} bdv_conv_geom;', hdr, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in re.findall(r'int32_t\s+([^;]+);', body):
        fields += [n.strip() for n in decl.split(',')]
    assert fields == [n for n, _ in bd._lib.ConvGeom._fields_]
    assert all(t is ctypes.c_int32 for _, t in bd._lib.ConvGeom._fields_)
    assert ctypes.sizeof(bd._lib.ConvGeom) == 4 * len(fields)


def test_stale_library_is_refused(monkeypatch):
    """The binding loads only a library built from the sources next to it (bdv_source_hash == sha256 of csrc + header)."""
    L = bd._lib
    assert L.lib().bdv_source_hash().decode() == L.source_hash()
    mk = open(os.path.join(os.path.dirname(L.LIB_PATH), 'Makefile')).read()
    hashed = re.search(r'^SRCS = (.*)$', mk, re.M).group(1).split() + re.search(r'^HASHED = \$\(SRCS\) (.*)$', mk, re.M).group(1).split()
    assert tuple(hashed) == L.HASHED_SOURCES
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'source_hash', lambda: '0' * 64)
    with pytest.raises(L.HipExtensionError, match='stale library'):
        L.lib()


def test_registry_contract():
    assert 'CILRecognizer2D' in bd.RECOGNIZERS and 'ResNetTSM' in bd.BACKBONES and 'IncrementalTSMHead' in bd.HEADS
    assert 'LSCLoss' in bd.LOSSES and 'ACMSmoothCE' in bd.LOSSES
    assert 'CILTSMOptimizerConstructorImprovised' in bd.OPTIMIZER_BUILDERS
    with pytest.raises(KeyError):
        bd.build_model(dict(type='NoSuchRecognizer'))
    with pytest.raises(KeyError):
        bd.build_loss(dict(type='NoSuchLoss'))
    loss = bd.build_loss(dict(type='LSCLoss'))
    assert isinstance(loss.eta, nn.Parameter) and loss.learnable_eta and loss.margin == 0.6

    @bd.LOSSES.register_module()
    class _MyLoss(nn.Module):
        pass
    assert isinstance(bd.build_loss(dict(type='_MyLoss')), _MyLoss)


@pytest.mark.parametrize('depth', [18, 34, 50])
def test_state_dict_layout_matches_oracle_and_roundtrips(depth):
    cfg = _cfg(depth)
    m, o = bd.build_model(copy.deepcopy(cfg)), O.build_model(copy.deepcopy(cfg))
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm.keys()) == list(so.keys())
    assert all(sm[k].shape == so[k].shape for k in sm)
    m.load_state_dict(so)
    assert all(torch.equal(m.state_dict()[k], so[k]) for k in so)
    # conv weights live in channels_last storage = the kernels' [Cout][R][S][Cin] layout
    w = m.backbone.layer1[0].conv2.conv.weight if depth == 50 else m.backbone.layer1[0].conv1.conv.net.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()
    # update_fc grows the classifier and keeps old rows; prev_model.load_state_dict(current) round-trips
    old = m.cls_head.fc_cls.weights.detach().clone()
    m.update_fc(12)
    assert m.cls_head.num_classes == 12 and torch.equal(m.cls_head.fc_cls.weights.detach()[:7], old)
    prev = bd.build_model(copy.deepcopy(cfg))
    prev.update_fc(12)
    prev.load_state_dict(m.state_dict())


def test_plugin_surface_attributes():
    m = bd.build_model(_cfg(50))
    assert m.cls_head.num_segments == 8 and callable(m.cls_head.consensus) and m.test_cfg['average_clips'] == 'prob'
    m.test_cfg['average_clips'] = 'score'
    for name in ['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool']:
        assert isinstance(bd.rgetattr(m, name), nn.Module)
    with pytest.raises(AttributeError):
        bd.OutputHook(m, ['backbone.layer9'])
    hook = bd.OutputHook(m, ['backbone.layer4'])
    pickle.loads(pickle.dumps(hook.tap('backbone.layer4')))      # picklable wrapper (ddp_spawn)
    m.freeze_backbone()
    assert not any(p.requires_grad for p in m.backbone.parameters()) and m.cls_head.fc_cls.weights.requires_grad
    m.unfreeze_backbone()
    assert all(p.requires_grad for p in m.backbone.parameters())
    with pytest.raises(ValueError):
        m(torch.zeros(1, 8, 3, 8, 8))                                       # label required when return_loss
    pickle.loads(pickle.dumps(m))                                           # whole model pickles (no ctypes handle inside)


def test_optimizer_constructor_groups_match_reference_semantics():
    opt_cfg = dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0),
                   lr=0.01, momentum=0.9, weight_decay=0.0001)
    m = bd.build_model(_cfg(50))
    opt = bd.build_optimizer(m, opt_cfg)
    got = [(len(g['params']), g['lr'], g['weight_decay']) for g in opt.param_groups]
    assert got == [(1, 0.01, 1e-4), (52, 0.01, 1e-4), (106, 0.01, 0), (2, 0.05, 1e-4)]
    ref_groups = [g for g in O.param_groups(O.build_model(_cfg(50)), 0.01, 1e-4, 5.0) if g['params']]
    assert got == [(len(g['params']), g['lr'], g['weight_decay']) for g in ref_groups]
    m2 = bd.build_model(_cfg(18, 'SimpleLinear', 'CrossEntropyLoss'))
    got2 = [(len(g['params']), g['lr'], g['weight_decay']) for g in bd.build_optimizer(m2, opt_cfg).param_groups]
    assert got2 == [(1, 0.01, 1e-4), (19, 0.01, 1e-4), (40, 0.01, 0), (1, 0.05, 1e-4), (1, 0.1, 0)]
    # the eta parameter sits in the classifier group; a frozen eta is left out
    assert any(p is m.cls_head.loss_cls.eta for p in opt.param_groups[-1]['params'])
    # unknown parameter-owning leaf -> ValueError, exactly like tsm.py:268-271
    class Odd(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(3))
    m.cls_head.odd = Odd()
    with pytest.raises(ValueError, match='New atomic module type'):
        bd.build_optimizer(m, opt_cfg)
    # the non-Improvised constructor rejects IncrementalNet (unknown leaf there) and uses 0.2x for fc_lr5
    with pytest.raises(ValueError):
        bd.build_optimizer(m2, dict(opt_cfg, constructor='CILTSMOptimizerConstructor', paramwise_cfg=dict(fc_lr5=True)))
    m3 = bd.build_model(_cfg(18))
    o3 = bd.build_optimizer(m3, dict(opt_cfg, constructor='CILTSMOptimizerConstructor', paramwise_cfg=dict(fc_lr5=True)))
    assert abs(o3.param_groups[-1]['lr'] - 0.002) < 1e-12
    # LR schedulers drive the fused optimizer like any torch optimizer
    sched = bd.build_lr_scheduler(opt := bd.build_optimizer(bd.build_model(_cfg(18)), opt_cfg),
                                  dict(type='MultiStepLR', params=dict(milestones=[1, 2], gamma=0.1)))
    sched.step()
    assert abs(opt.param_groups[0]['lr'] - 0.001) < 1e-12 and abs(opt.param_groups[-1]['lr'] - 0.005) < 1e-12


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing somewhere else."""
    m = bd.build_model(_cfg(18))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 8, 3, 32, 32), torch.zeros(1, 1, dtype=torch.long))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        bd.LSCLoss()(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))
    fe = bd.BackgroundMixFrontEnd()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        fe(torch.zeros(1, 8, 4, 4, 3, dtype=torch.uint8))
    # and the product package never imports the oracle
    import sys
    src_dir = os.path.join(ROOT, 'background-debiased-video-cil_amd')
    for fn in os.listdir(src_dir):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(src_dir, fn)).read(), fn


def test_unsupported_configs_raise():
    cfg = _cfg(18)
    cfg['backbone']['shift_place'] = 'block'
    with pytest.raises(NotImplementedError):
        bd.build_model(cfg)
    cfg = _cfg(18)
    cfg['backbone']['depth'] = 101
    with pytest.raises(KeyError):
        bd.build_model(cfg)
    cfg = _cfg(18)
    cfg['backbone']['pretrained'] = 'https://download.pytorch.org/models/resnet18-f37072fd.pth'
    with pytest.raises(FileNotFoundError):
        bd.build_model(cfg)
    m = bd.build_model(_cfg(18))
    m.cls_head.fc_cls = nn.Linear(4, 4)
    with pytest.raises(ValueError, match='init_weights'):
        m.update_fc(9)
