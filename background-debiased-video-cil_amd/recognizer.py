"""Recognizer plugin surface: UPSTREAM BaseRecognizer/Recognizer2D semantics and the
reference's ``CILRecognizer2D`` (libs/models/base.py:8-42)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import RECOGNIZERS, build_backbone, build_head
from .resnet_tsm import Nhwc4Frames


class Recognizer2D(nn.Module):
    def __init__(self, backbone, cls_head=None, neck=None, train_cfg=None, test_cfg=None):
        super().__init__()
        if neck is not None:
            raise NotImplementedError('necks are not used by any CIL config')
        self.backbone = build_backbone(backbone)
        self.cls_head = build_head(cls_head) if cls_head else None
        self.train_cfg = train_cfg
        self.test_cfg = dict(test_cfg) if test_cfg else {}
        self.init_weights()

    @property
    def with_cls_head(self):
        return self.cls_head is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_cls_head:
            self.cls_head.init_weights()

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    @staticmethod
    def _frames(imgs):
        """(B, T, 3, H, W) -> ((B*T, 3, H, W) or Nhwc4Frames, B, num_segs)."""
        if isinstance(imgs, Nhwc4Frames):
            return imgs, imgs.batches, imgs.num_segments
        batches = imgs.shape[0]
        x = imgs.reshape((-1,) + imgs.shape[2:])
        return x, batches, x.shape[0] // batches

    def forward_train(self, imgs, labels, **kwargs):
        x, batches, num_segs = self._frames(imgs)
        feat = self.backbone(x)
        cls_score = self.cls_head(feat, num_segs)
        gt_labels = labels.squeeze()
        return dict(self.cls_head.loss(cls_score, gt_labels, **kwargs))

    def _do_test(self, imgs):
        x, batches, num_segs = self._frames(imgs)
        feat = self.backbone(x)
        cls_score = self.cls_head(feat, num_segs)
        if cls_score.size(0) % batches != 0:
            raise ValueError('cls_score rows not divisible by the batch size')
        return self.average_clip(cls_score, cls_score.size(0) // batches)

    def average_clip(self, cls_score, num_segs=1):
        mode = self.test_cfg.get('average_clips', None)
        if mode not in ['score', 'prob', None]:
            raise ValueError(f'{mode} is not supported. Currently supported ones are ["score", "prob", None]')
        if mode is None:
            return cls_score
        batches = cls_score.shape[0] // num_segs
        if mode == 'score':
            # differentiable (ICARLModel.training_step back-propagates through forward_test, icarl.py:100)
            return Fn.ConsensusFn.apply(cls_score.view(batches, num_segs, -1)).squeeze(1)
        if torch.is_grad_enabled() and cls_score.requires_grad:
            raise NotImplementedError("average_clips='prob' is an inference-only path; use 'score' to back-propagate")
        return K.softmax_mean(cls_score.contiguous(), batches, num_segs, apply_softmax=True)


@RECOGNIZERS.register_module()
class CILRecognizer2D(Recognizer2D):
    """libs/models/base.py:8-42."""

    def forward_train(self, imgs, labels, **kwargs):
        return super().forward_train(imgs, labels, num_classes=self.cls_head.num_classes, **kwargs)

    def forward_test(self, imgs):
        if self.test_cfg.get('fcn_test', False):
            raise NotImplementedError('fcn_test is not used by any CIL config')
        return self._do_test(imgs)

    def update_fc(self, nb_classes):
        self.cls_head.update_fc(nb_classes)

    def freeze_backbone(self):
        for param in self.backbone.parameters():
            param.requires_grad = False

    def unfreeze_backbone(self):
        for param in self.backbone.parameters():
            param.requires_grad = True
