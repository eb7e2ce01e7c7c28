"""``libs.losses`` API on the HIP kernels: same class names, constructor arguments and call signatures
as libs/losses/lsc_loss.py and libs/losses/acm_smooth_ce.py, registered in ``LOSSES``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import LOSSES


def _class_weights(cw, like: torch.Tensor):
    """``class_weights`` as the reference accepts it (a tensor indexed by the targets, or anything ``F.cross_entropy`` takes as
    ``weight``) -> contiguous fp32 (K,) on the scores' device, or None."""
    if cw is None:
        return None
    cw = torch.as_tensor(cw, dtype=torch.float32).to(like.device).contiguous()
    if cw.dim() != 1 or cw.numel() != like.shape[1]:
        raise ValueError(f'class_weights: expected {like.shape[1]} values (one per class), got shape {tuple(cw.shape)}')
    return cw


def _weighted_ce(score, labels, cw):
    """``F.cross_entropy(score, labels, weight=cw, reduction='mean')`` = sum_i w[y_i] * ce_i / sum_i w[y_i]: the soft-target kernel
    on one-hot rows scaled by B * w[y_i] / sum_j w[y_j] (its mean over the batch is then the weighted mean)."""
    labels = labels.contiguous()
    if cw is None:
        return Fn.SoftCEFn.apply(score, None, labels)
    w = cw[labels]
    rows = torch.zeros_like(score)
    rows.scatter_(1, labels.view(-1, 1), (w * (labels.numel() / w.sum())).view(-1, 1))     # (B, K)-sized plumbing
    return Fn.SoftCEFn.apply(score, rows, None)


@LOSSES.register_module()
class LSCLoss(nn.Module):
    """libs/losses/lsc_loss.py:8-58.  ``forward(similarities (B,K), targets (B,), **kwargs) -> scalar``."""

    def __init__(self, eta=1.0, margin=0.6, learnable_eta=True, exclude_pos_denominator=True, hinge_proxynca=True,
                 class_weights=None):
        super().__init__()
        self.margin = margin
        self.exclude_pos_denominator = exclude_pos_denominator
        self.hinge_proxynca = hinge_proxynca
        self.class_weights = class_weights
        self.learnable_eta = learnable_eta
        self.eta = nn.Parameter(torch.Tensor([eta]), requires_grad=self.learnable_eta)

    def forward(self, similarities: torch.Tensor, targets: torch.Tensor, **kwargs):
        if targets.dim() == 0:
            targets = targets.unsqueeze(0)
        cw = _class_weights(self.class_weights, similarities)
        if self.exclude_pos_denominator:
            return Fn.LSCLossFn.apply(similarities, targets, self.eta, float(self.margin), bool(self.hinge_proxynca), cw)
        return _weighted_ce(similarities, targets, cw)                          # lsc_loss.py:58: (weighted) cross entropy


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Mean cross-entropy on integer labels (BASELINE config 2 "CE loss only").  Unlike UPSTREAM mmaction
    CrossEntropyLoss it tolerates the ``batch_data`` / ``num_classes`` kwargs CILRecognizer2D adds
    (SURVEY Appendix A notes the upstream one would raise on them)."""

    def __init__(self, loss_weight=1.0, class_weight=None):
        super().__init__()
        self.loss_weight = loss_weight
        self.class_weight = None if class_weight is None else torch.as_tensor(class_weight, dtype=torch.float32)   # UPSTREAM: a list

    def forward(self, cls_score, label, **kwargs):
        if label.dim() == 0:
            label = label.unsqueeze(0)
        loss = _weighted_ce(cls_score, label, _class_weights(self.class_weight, cls_score))
        return loss if self.loss_weight == 1.0 else loss * self.loss_weight


@LOSSES.register_module()
class SoftTargetCrossEntropy(nn.Module):
    """Loss of ``ICARLModel.training_step`` (libs/cil/icarl.py:101-125): one-hot targets whose old-class
    rows are replaced by softmax(prev-model logits); ``mean_b(-sum_k tgt * log_softmax(score))``.  ``base_targets``
    (B, K) replaces the one-hot rows (the foreground-ratio soft labels of icarl.py:103-111)."""

    def forward(self, cls_score, labels, prev_logits=None, prev_num_classes=0, base_targets=None, **kwargs):
        tgt = K.icarl_targets(labels.reshape(-1).contiguous(), None if prev_logits is None else prev_logits.contiguous(),
                              prev_num_classes, cls_score.shape[1], base_targets)
        return Fn.SoftCEFn.apply(cls_score, tgt, None)


@LOSSES.register_module()
class ACMSmoothCE(nn.Module):
    """ActorCutMix smooth-label loss (libs/losses/acm_smooth_ce.py:8-30): labels are mixed with the background label by
    ``lambda = 1 - (1 - foreground_ratio)^alpha`` and the loss is ``mean_b(sum_k y * log_softmax(score))`` -- with the
    reference's sign (no negation).  No shipped config reaches it (all ACM configs train with methods='icarl', which
    bypasses ``head.loss``); it is provided so that the ``libs.losses`` plugin surface is complete.  Like the reference,
    ``batch_data['background_label']`` is modified in place (-1 -> 0)."""

    def __init__(self, alpha: float = 4):
        super().__init__()
        self.alpha = alpha

    def forward(self, cls_score, labels, batch_data, num_classes, **kwargs):
        background_labels = torch.squeeze(batch_data['background_label'], dim=1)
        background_labels[background_labels == -1] = 0
        fg = batch_data['foreground_ratio'].reshape(-1).to(torch.float32).contiguous()
        tgt = K.acm_targets(labels.reshape(-1).contiguous(), background_labels.contiguous(), fg, self.alpha, int(num_classes))
        return -Fn.SoftCEFn.apply(cls_score, tgt, None)
