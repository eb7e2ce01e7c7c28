"""Parameter-group construction and the fused multi-tensor SGD step.

* ``CILTSMOptimizerConstructorImprovised`` / ``CILTSMOptimizerConstructor`` reproduce the grouping of
  libs/models/cil_heads/tsm.py:68-303 (including the ``ValueError`` on unknown parameter-owning leaf modules and
  the 0.2x quirk of the non-Improvised variant).
* ``FusedSGD`` is a ``torch.optim.Optimizer`` (so ``MultiStepLR`` etc. drive ``group['lr']`` as usual) whose
  ``step`` is one multi-tensor HIP launch; ``clip_grad_norm_`` computes the global-norm clip coefficient on the
  device (no host sync), the way PL's ``gradient_clip_val`` does for task > 0 (libs/cil/cil.py:743).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm
from torch.nn.modules.conv import _ConvNd

from . import kernels as K
from .functional import join_side_stream, side_stream, side_stream_enabled
from .heads import LSC, IncrementalNet
from .losses import LSCLoss
from .registry import OPTIMIZER_BUILDERS, build_from_cfg


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same element order in memory: strides agree on every dimension longer than 1 (a 1x1 conv weight in
    channels_last order and its plainly contiguous parameter differ only in the strides of the size-1 dims)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


PREFETCH_PLANES = os.environ.get('BDVCIL_PREFETCH_PLANES', '1') != '0'


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.0, weight_decay=0.0, dampening=0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError('dampening / nesterov are not used by the reference (configs: SGD momentum 0.9)')
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._tables = None
        self._grad_scale = 1.0
        self._clip_active = False
        self._sqnorm = None
        self._coef = None

    # -- table management ---------------------------------------------------------------
    def _active(self):
        out = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is not None:
                    out.append((gi, p))
        return out

    def _build_tables(self, active):
        dev = active[0][1].device
        for _, p in active:
            if not p.is_cuda:
                raise RuntimeError('FusedSGD runs on the GPU only (no CPU fallback)')
            g = p.grad
            if g.dtype != torch.float32 or p.dtype != torch.float32:
                raise TypeError('FusedSGD: fp32 only')
            if not _same_layout(g, p) or not K._dense_storage(p).is_contiguous():
                p.grad = torch.empty_like(p).copy_(g)          # rare: make the layouts agree
            st = self.state[p]
            if 'momentum_buffer' not in st:
                st['momentum_buffer'] = torch.zeros_like(p)    # buf = g on the first step == 0.9*0 + g
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['momentum_buffer'].data_ptr()) for _, p in active)
        hyper = tuple((self.param_groups[gi]['lr'], self.param_groups[gi]['weight_decay']) for gi, _ in active)
        t = self._tables
        n = len(active)
        if t is None or t['count'] != n or t['dev'] != dev:
            table = torch.empty(4 * n, dtype=torch.int64, device=dev)
            t = dict(key=None, hyper=None, count=n, dev=dev, table=table, p=table[:n], g=table[n:2 * n], b=table[2 * n:3 * n],
                     n=table[3 * n:], staging=[torch.empty(4 * n, dtype=torch.int64).pin_memory() for _ in range(4)],
                     events=[None] * 4, slot=0)
            self._tables = t
        if t['key'] != key:
            # Gradients are fresh allocations every step (zero_grad(set_to_none=True)), so the pointer table changes every
            # step.  It goes to the device through a ring of pinned staging buffers with an asynchronous copy: a pageable
            # copy would make the host wait here for the whole backward pass.
            i = t['slot']
            t['slot'] = (i + 1) % len(t['staging'])
            if t['events'][i] is not None:
                t['events'][i].synchronize()                   # the copy issued from this buffer four rebuilds ago
            st = t['staging'][i]
            st[:3 * n].view(3, n).copy_(torch.tensor(key, dtype=torch.int64).t())
            st[3 * n:].copy_(torch.tensor([p.numel() for _, p in active], dtype=torch.int64))
            t['table'].copy_(st, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            t['events'][i] = ev
            t['key'] = key
        if t['hyper'] != hyper:
            t['hyper'] = hyper
            t['lr'] = torch.tensor([h[0] for h in hyper], dtype=torch.float32, device=dev)
            t['wd'] = torch.tensor([h[1] for h in hyper], dtype=torch.float32, device=dev)
        if self._sqnorm is None or self._sqnorm.device != dev:
            self._sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._coef = torch.ones(1, dtype=torch.float32, device=dev)
        return t

    # -- public API ---------------------------------------------------------------------
    def set_grad_scale(self, scale: float):
        """Multiplier applied to every gradient inside the step (1/world_size after a SUM all-reduce)."""
        self._grad_scale = float(scale)

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float):
        """Global L2-norm clip of all gradients; the coefficient stays on the device and is consumed by the next
        ``step``.  Returns the (device) total norm tensor."""
        active = self._active()
        if not active:
            return None
        join_side_stream()
        t = self._build_tables(active)
        K.multi_sqnorm(t['g'], t['n'], len(active), self._sqnorm)
        K.clip_coef(self._sqnorm, self._grad_scale, float(max_norm), self._coef)
        self._clip_active = True
        return self._sqnorm.sqrt() * self._grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active = self._active()
        if not active:
            return loss
        momenta = {g['momentum'] for g in self.param_groups}
        if len(momenta) != 1:
            raise NotImplementedError('FusedSGD: one momentum value for all groups')
        join_side_stream()
        t = self._build_tables(active)
        K.multi_sgd(t['p'], t['g'], t['b'], t['n'], t['lr'], t['wd'], len(active), momenta.pop(), self._grad_scale,
                    self._coef if self._clip_active else None)
        K.bump_weight_epoch([p for _, p in active])   # the kernel wrote these weights through raw pointers: their cached bf16 planes are stale
        if PREFETCH_PLANES and side_stream_enabled():
            # re-split them now, beside whatever the main stream does until the first conv of the next step needs them
            dev = active[0][1].device
            if dev.type == 'cuda':
                K.refresh_weight_planes(side_stream(dev))
        self._clip_active = False
        return loss


def _collect_groups(model: nn.Module, improvised: bool, fc_lr5: bool):
    first_w, first_b, normal_w, normal_b, lr5_w, lr10_b, bn = [], [], [], [], [], [], []
    conv_cnt = 0
    for m in model.modules():
        if isinstance(m, _ConvNd):
            ps = list(m.parameters())
            conv_cnt += 1
            (first_w if conv_cnt == 1 else normal_w).append(ps[0])
            if len(ps) == 2:
                (first_b if conv_cnt == 1 else normal_b).append(ps[1])
        elif isinstance(m, nn.Linear):
            ps = list(m.parameters())
            normal_w.append(ps[0])
            if len(ps) == 2:
                normal_b.append(ps[1])
        elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
            bn.extend(p for p in m.parameters() if p.requires_grad)
        elif isinstance(m, LSC):
            (lr5_w if (improvised or fc_lr5) else normal_w).append(list(m.parameters())[0])
        elif isinstance(m, LSCLoss):
            if m.learnable_eta:
                (lr5_w if (improvised or fc_lr5) else normal_w).append(list(m.parameters())[0])
        elif improvised and isinstance(m, IncrementalNet):
            ps = list(m.parameters())
            lr5_w.append(ps[0])
            lr10_b.append(ps[1])
        elif len(m._modules) == 0 and len(list(m.parameters())) > 0:
            raise ValueError(f'New atomic module type: {type(m)}. Need to give it a learning policy')
    return first_w, first_b, normal_w, normal_b, bn, lr5_w, lr10_b


class _ConstructorBase:
    improvised = True

    def __init__(self, optimizer_cfg: dict, paramwise_cfg: Optional[dict] = None):
        if not isinstance(optimizer_cfg, dict):
            raise TypeError('optimizer_cfg should be a dict')
        self.optimizer_cfg = dict(optimizer_cfg)
        self.paramwise_cfg = {} if paramwise_cfg is None else dict(paramwise_cfg)
        self.base_lr = self.optimizer_cfg.get('lr', None)
        self.base_wd = self.optimizer_cfg.get('weight_decay', None)

    def _multipliers(self):
        raise NotImplementedError

    def add_params(self, params: List[dict], model: nn.Module):
        fc_lr5 = bool(self.paramwise_cfg.get('fc_lr5', False))
        first_w, first_b, normal_w, normal_b, bn, lr5_w, lr10_b = _collect_groups(model, self.improvised, fc_lr5)
        m5, m10 = self._multipliers()
        params.append({'params': first_w, 'lr': self.base_lr, 'weight_decay': self.base_wd})
        params.append({'params': first_b, 'lr': self.base_lr * 2, 'weight_decay': 0})
        params.append({'params': normal_w, 'lr': self.base_lr, 'weight_decay': self.base_wd})
        params.append({'params': normal_b, 'lr': self.base_lr * 2, 'weight_decay': 0})
        params.append({'params': bn, 'lr': self.base_lr, 'weight_decay': 0})
        params.append({'params': lr5_w, 'lr': self.base_lr * m5, 'weight_decay': self.base_wd})
        params.append({'params': lr10_b, 'lr': self.base_lr * m10, 'weight_decay': 0})

    def __call__(self, model: nn.Module):
        if hasattr(model, 'module'):
            model = model.module
        cfg = dict(self.optimizer_cfg)
        typ = cfg.pop('type')
        if typ != 'SGD':
            raise KeyError(f'optimizer type {typ!r}: only SGD is on the HIP path (all CIL configs use SGD)')
        params: List[dict] = []
        self.add_params(params, model)
        params = [g for g in params if len(g['params']) > 0]
        return FusedSGD(params, **cfg)


@OPTIMIZER_BUILDERS.register_module()
class CILTSMOptimizerConstructorImprovised(_ConstructorBase):
    """libs/models/cil_heads/tsm.py:190-303."""
    improvised = True

    def _multipliers(self):
        f = self.paramwise_cfg['fc_lr_scale_factor']
        return f, f * 2


@OPTIMIZER_BUILDERS.register_module()
class CILTSMOptimizerConstructor(_ConstructorBase):
    """libs/models/cil_heads/tsm.py:68-186 (lr5 group gets 0.2x lr -- Appendix C.9 -- and IncrementalNet is an
    unknown leaf there, so it raises exactly like the reference)."""
    improvised = False

    def _multipliers(self):
        return 0.2, 10


def build_optimizer(model: nn.Module, cfg: dict):
    """mmcv ``build_optimizer(model, cfg)`` as used at libs/cil/cil.py:467."""
    optimizer_cfg = dict(cfg)
    constructor_type = optimizer_cfg.pop('constructor', 'DefaultOptimizerConstructor')
    paramwise_cfg = optimizer_cfg.pop('paramwise_cfg', None)
    if constructor_type == 'DefaultOptimizerConstructor':
        typ = optimizer_cfg.pop('type')
        if typ != 'SGD':
            raise KeyError(f'optimizer type {typ!r}: only SGD is on the HIP path')
        return FusedSGD([p for p in model.parameters() if p.requires_grad], **optimizer_cfg)
    ctor = build_from_cfg(dict(type=constructor_type, optimizer_cfg=optimizer_cfg, paramwise_cfg=paramwise_cfg),
                          OPTIMIZER_BUILDERS)
    return ctor(model)


def build_lr_scheduler(optimizer, lr_scheduler_config: dict):
    """libs/utils.py build_lr_scheduler: a torch scheduler looked up by name."""
    cls = getattr(torch.optim.lr_scheduler, lr_scheduler_config['type'])
    return cls(optimizer, **lr_scheduler_config.get('params', {}))
