"""Incremental TSM classification head on the HIP kernels.

Same classes / attributes as libs/models/cil_heads/{tsm,cosine_linear,inc_net}.py and UPSTREAM mmaction
``TSMHead`` / ``BaseHead`` / ``AvgConsensus``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import HEADS, build_loss


class LSC(nn.Module):
    """Local Similarity Classifier, libs/models/cil_heads/cosine_linear.py:6-55."""

    def __init__(self, in_features: int, out_features: int, nb_proxies: int = 3):
        super().__init__()
        self.in_features, self.out_features, self.nb_proxies = in_features, out_features, nb_proxies
        self.weights = nn.Parameter(torch.empty(out_features, self.nb_proxies * in_features), requires_grad=True)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_normal_(self.weights, nonlinearity='linear')

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.LSCFn.apply(x, self.weights, self.out_features, self.nb_proxies)

    def update_fc(self, nb_classes):
        new_weight = torch.empty(nb_classes, self.nb_proxies * self.in_features).type_as(self.weights.data)
        nn.init.kaiming_normal_(new_weight, nonlinearity='linear')
        new_weight[:self.out_features] = self.weights.data
        self.weights = nn.Parameter(new_weight, requires_grad=True)
        self.out_features = nb_classes

    def __repr__(self):
        return 'LocalSimilarityClassifier(in_features: {}, out_features: {}, nb_proxies: {})'.format(
            self.in_features, self.out_features, self.nb_proxies)


class IncrementalNet(nn.Module):
    """Growable linear layer, libs/models/cil_heads/inc_net.py:6-37."""

    def __init__(self, in_features: int, out_features: int, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.Tensor(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, nonlinearity='linear')
        nn.init.constant_(self.bias, 0)

    def update_fc(self, nb_classes):
        new_weight = torch.empty(nb_classes, self.in_features).type_as(self.weight.data)
        nn.init.kaiming_normal_(new_weight, nonlinearity='linear')
        new_weight[:self.out_features] = self.weight.data
        self.weight = nn.Parameter(new_weight, requires_grad=True)
        new_bias = torch.empty(nb_classes).type_as(self.bias.data)
        nn.init.constant_(new_bias, 0)
        new_bias[:self.out_features] = self.bias.data
        self.bias = nn.Parameter(new_bias, requires_grad=True)
        self.out_features = nb_classes

    def forward(self, x):
        return Fn.LinearFn.apply(x, self.weight, self.bias)


inc_linear_layers = {'SimpleLinear': IncrementalNet, 'LocalSimilarityClassifier': LSC}


class AvgConsensus(nn.Module):
    """UPSTREAM AvgConsensus: mean over ``dim`` with keepdim."""

    def __init__(self, dim=1):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        if self.dim != 1 or x.dim() != 3:
            raise NotImplementedError('AvgConsensus: only dim=1 on (B, T, K) is on the HIP path')
        return Fn.ConsensusFn.apply(x)


class AvgPool2dTo1(nn.Module):
    """``cls_head.avg_pool`` (AdaptiveAvgPool2d(1)); hooked for KD and NME representations."""

    def forward(self, x):
        return Fn.AvgPoolFn.apply(x)


class HipDropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self._calls = 0

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        self._calls += 1
        seed = (torch.initial_seed() * 1000003 + self._calls) & 0x7FFFFFFFFFFFFFFF
        return Fn.DropoutFn.apply(x, float(self.p), seed)


@HEADS.register_module()
class IncrementalTSMHead(nn.Module):
    """libs/models/cil_heads/tsm.py:21-64 on top of UPSTREAM TSMHead/BaseHead semantics."""

    def __init__(self, num_classes, in_channels, inc_head_config=dict(type='LocalSimilarityClassifier'), num_segments=8,
                 loss_cls=dict(type='CrossEntropyLoss'), spatial_type='avg', consensus=dict(type='AvgConsensus', dim=1),
                 dropout_ratio=0.8, init_std=0.001, is_shift=True, temporal_pool=False, multi_class=False,
                 label_smooth_eps=0.0, topk=(1, 5), **kwargs):
        super().__init__()
        if temporal_pool or multi_class or label_smooth_eps:
            raise NotImplementedError('temporal_pool / multi_class / label smoothing are not used by any CIL config')
        if spatial_type != 'avg':
            raise NotImplementedError("only spatial_type='avg'")
        self.num_classes, self.in_channels, self.num_segments = num_classes, in_channels, num_segments
        self.dropout_ratio, self.init_std, self.is_shift, self.temporal_pool = dropout_ratio, init_std, is_shift, temporal_pool
        self.multi_class, self.label_smooth_eps, self.topk = multi_class, label_smooth_eps, tuple(topk)
        self.loss_cls = build_loss(loss_cls)
        consensus_ = dict(consensus)
        if consensus_.pop('type') != 'AvgConsensus':
            raise KeyError('only AvgConsensus is supported')
        self.consensus = AvgConsensus(**consensus_)
        self.dropout = HipDropout(self.dropout_ratio) if self.dropout_ratio != 0 else None
        self.fc_cls = nn.Linear(self.in_channels, self.num_classes)       # replaced in init_weights (tsm.py:51-56)
        self.avg_pool = AvgPool2dTo1()
        self.inc_head_config = dict(inc_head_config)
        self.inc_head_config['in_features'] = in_channels

    def init_weights(self):
        cfg = self.inc_head_config.copy()
        head_type = inc_linear_layers[cfg.pop('type')]
        self.fc_cls = head_type(**cfg)
        self.fc_cls.update_fc(self.num_classes)

    def update_fc(self, nb_classes):
        if not hasattr(self.fc_cls, 'update_fc'):
            raise ValueError('Replace fc layer with incremental fc layer with "init_weights" method '
                             'before using "update_fc" method')
        dev = next(self.fc_cls.parameters()).device
        self.fc_cls.update_fc(nb_classes)
        self.fc_cls.to(dev)
        self.num_classes = nb_classes

    def forward(self, x, num_segs=None):
        """x: (N, C, h, w) -> (B, K).  Views by ``self.num_segments`` like UPSTREAM TSMHead."""
        x = self.avg_pool(x)
        x = torch.flatten(x, 1)
        if self.dropout is not None:
            x = self.dropout(x)
        cls_score = self.fc_cls(x)
        cls_score = cls_score.view((-1, self.num_segments) + cls_score.size()[1:])
        cls_score = self.consensus(cls_score)
        return cls_score.squeeze(1)

    def loss(self, cls_score, labels, **kwargs):
        """UPSTREAM BaseHead.loss; top-k accuracy is computed on the device (no D2H sync per step)."""
        losses = dict()
        if labels.shape == torch.Size([]):
            labels = labels.unsqueeze(0)
        if cls_score.size() != labels.size():
            acc = K.topk_acc(cls_score.detach().contiguous(), labels.contiguous())
            for i, k in enumerate(self.topk[:2]):
                losses[f'top{k}_acc'] = acc[i]
        loss_cls = self.loss_cls(cls_score, labels, **kwargs)
        if isinstance(loss_cls, dict):
            losses.update(loss_cls)
        else:
            losses['loss_cls'] = loss_cls
        return losses
