"""Incremental TSM classification head on the HIP kernels.

Same classes / attributes as libs/models/cil_heads/{tsm,cosine_linear,inc_net}.py and UPSTREAM mmaction
``TSMHead`` / ``BaseHead`` / ``AvgConsensus``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import HEADS, build_loss


def _grown(old: torch.Tensor, rows: int) -> torch.Tensor:
    """A (rows, ...) tensor like ``old`` whose first ``len(old)`` rows are ``old`` and whose other rows are freshly drawn
    kaiming-normal values (the whole tensor is drawn first, as the reference does, so the random stream is consumed
    identically: cosine_linear.py:45-50, inc_net.py:23-34)."""
    out = old.new_empty((rows,) + tuple(old.shape[1:]))
    if out.dim() > 1:
        nn.init.kaiming_normal_(out, nonlinearity='linear')
    else:
        out.zero_()
    out[:old.shape[0]].copy_(old)
    return out


class LSC(nn.Module):
    """Local Similarity Classifier (plugin surface of libs/models/cil_heads/cosine_linear.py:6-55): ``weights`` is
    (out_features, nb_proxies * in_features); the forward is the fused HIP kernel."""

    def __init__(self, in_features: int, out_features: int, nb_proxies: int = 3):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.nb_proxies = nb_proxies
        self.weights = nn.Parameter(torch.empty(out_features, nb_proxies * in_features))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_normal_(self.weights, nonlinearity='linear')

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.LSCFn.apply(x, self.weights, self.out_features, self.nb_proxies)

    def update_fc(self, nb_classes):
        """Grow to ``nb_classes`` rows keeping the old ones; ``weights`` becomes a NEW Parameter object (the optimizer is
        rebuilt per task)."""
        self.weights = nn.Parameter(_grown(self.weights.data, nb_classes), requires_grad=True)
        self.out_features = nb_classes

    def __repr__(self):
        return (f'LocalSimilarityClassifier(in_features: {self.in_features}, out_features: {self.out_features}, '
                f'nb_proxies: {self.nb_proxies})')


class IncrementalNet(nn.Module):
    """Growable linear layer (plugin surface of libs/models/cil_heads/inc_net.py:6-37)."""

    def __init__(self, in_features: int, out_features: int, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, nonlinearity='linear')
        nn.init.constant_(self.bias, 0)             # like the reference, bias=False fails here

    def update_fc(self, nb_classes):
        self.weight = nn.Parameter(_grown(self.weight.data, nb_classes), requires_grad=True)
        self.bias = nn.Parameter(_grown(self.bias.data, nb_classes), requires_grad=True)     # new entries zero
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


_DROPOUT_DRAWS = [0]      # per process, shared by all HipDropout instances: rebuilding a model does not replay the masks


class HipDropout(nn.Module):
    """nn.Dropout with a counter-based generator in the kernel.  The 64-bit seed of a call mixes torch's initial seed, the
    rank of the process (ranks seeded alike still draw different masks) and the number of draws made so far."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        _DROPOUT_DRAWS[0] += 1
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        seed = ((torch.initial_seed() * 1000003 + rank) * 1000003 + _DROPOUT_DRAWS[0]) & 0x7FFFFFFFFFFFFFFF
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
