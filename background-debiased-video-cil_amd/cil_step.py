"""The CIL training-step arithmetic without Lightning: what ``BaseCIL.training_step`` (libs/cil/cil.py:512-556)
and ``ICARLModel.training_step`` (libs/cil/icarl.py:97-130) compute per batch, plus a small engine that runs
forward + backward + (RCCL all-reduce) + clip + fused SGD for one step.  The task loop, exemplar bookkeeping and
datasets stay with the caller (out of scope: SURVEY section 2 #12-#14)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .ddp import GradAllReducer
from .hooks import OutputHook
from .losses import SoftTargetCrossEntropy


def base_training_step(current_model: nn.Module, batch_data: Dict[str, torch.Tensor], current_task: int = 0,
                       prev_model: Optional[nn.Module] = None, current_hooks: Optional[OutputHook] = None,
                       prev_hooks: Optional[OutputHook] = None, kd_modules_names: Sequence[str] = (),
                       kd_weight_by_module: Sequence[float] = (), adaptive_scale_factors: Sequence[float] = (),
                       kd_exemplar_only: bool = False, previous_task_num_classes: int = 0) -> Dict[str, torch.Tensor]:
    """libs/cil/cil.py:512-556.  Returns the ``losses`` dict with an extra ``'loss'`` entry (the value the
    reference returns from ``training_step``)."""
    imgs, labels = batch_data['imgs'], batch_data['label']
    losses = current_model(imgs, labels, batch_data=batch_data)
    use_kd = prev_model is not None and len(kd_modules_names) > 0
    if use_kd and current_task > 0:
        total_kd_loss = 0
        prev_model.eval()
        with torch.no_grad():
            prev_model.forward_test(imgs)
        scale_factor = adaptive_scale_factors[current_task]
        for m_name, kd_weight in zip(kd_modules_names, kd_weight_by_module):
            cur = current_hooks.get_layer_output(m_name)
            prev = prev_hooks.get_layer_output(m_name).detach()
            if kd_exemplar_only:
                # cil.py:529-536: rows of the hooked tensors selected by the batch positions of the old-class samples.
                # (The first dimension of the backbone features is frames, not samples; the reference indexes it with
                # sample positions all the same, and so does this.)
                indices = (batch_data['label'].view(-1) < previous_task_num_classes).nonzero().reshape(-1)
                if indices.nelement():
                    kd_loss = Fn.kd_mse(cur.index_select(0, indices).contiguous(), prev.index_select(0, indices).contiguous())
                else:
                    kd_loss = 0
            else:
                kd_loss = Fn.kd_mse(cur, prev)
            losses[m_name] = kd_loss
            total_kd_loss = total_kd_loss + scale_factor * kd_weight * kd_loss
        losses['kd_loss'] = total_kd_loss
    else:
        losses['kd_loss'] = 0.
    loss = losses['kd_loss'] + losses['loss_cls']
    if 'loss_bg_mixed' in losses:
        loss = loss + losses['loss_bg_mixed']
    losses['loss'] = loss
    return losses


_soft_ce = SoftTargetCrossEntropy()


def icarl_training_step(current_model: nn.Module, batch_data: Dict[str, torch.Tensor], num_classes: int,
                        current_task: int = 0, prev_model: Optional[nn.Module] = None,
                        previous_task_num_classes: int = 0) -> torch.Tensor:
    """libs/cil/icarl.py:97-125, including the soft labels of the ``foreground_ratio`` branch (:103-111; like the
    reference it rewrites ``batch_data['background_label']`` -1 -> 0 in place).

    Deliberate deviation (parity unpinned, no fixture covers it): the teacher runs in EVAL mode (running BatchNorm
    statistics, no dropout) on the whole batch and only its old-class rows are used.  ``ICARLModel.training_step`` never
    calls ``prev_model.eval()`` (unlike ``BaseCIL`` at cil.py:520), so under Lightning's ``fit`` the reference's teacher
    runs in train mode on ``imgs[indices]`` alone: batch-statistics BatchNorm over the old-class clips of the batch,
    dropout 0.5 on its features, and running statistics that drift with every step.  A frozen teacher is what the
    method describes; with eval-mode BatchNorm the rows are independent, so evaluating the whole batch equals the gather."""
    imgs, targets = batch_data['imgs'], batch_data['label']
    cls_score = current_model(imgs, return_loss=False)
    base = None
    if 'foreground_ratio' in batch_data:
        background_labels = torch.squeeze(batch_data['background_label'], dim=1)
        background_labels[background_labels == -1] = 0
        fg = batch_data['foreground_ratio'].reshape(-1).to(torch.float32).contiguous()
        base = K.acm_targets(targets.reshape(-1).contiguous(), background_labels.contiguous(), fg, 4.0, num_classes)
    prev_logits = None
    if current_task > 0 and prev_model is not None:
        with torch.no_grad():
            prev_logits = prev_model(imgs, return_loss=False)
    return _soft_ce(cls_score, targets.view(-1), prev_logits=prev_logits, prev_num_classes=previous_task_num_classes,
                    base_targets=base)


def tubemix_draw(batch_size: int, rows: int, cols: int, alpha, prob: float):
    """The random decisions of ``tubemix`` / ``rand_bbox`` (libs/cil/icarl_video_mix.py:48-81) in the reference's order:
    ``random.random()``, ``torch.randperm``, ``np.random.beta``, two ``np.random.randint``.  Returns None (no mix) or
    ``(batch_idx, (r1, c1, r2, c2), lam)`` with the box over (rows, cols) = the last two dimensions of the clips and
    ``lam`` the area-corrected mixing weight.  (``np.int`` of the reference is spelled ``int`` here: it no longer exists
    in the numpy of this image.)"""
    import random

    import numpy as np
    if prob < 0:
        raise ValueError('prob must be a positive value')
    if not (random.random() > 1 - prob):
        return None
    batch_idx = torch.randperm(batch_size)
    lam = np.random.beta(alpha, alpha)
    cut_rat = np.sqrt(1. - lam)
    cut_r, cut_c = int(np.asarray(rows * cut_rat).reshape(-1)[0]), int(np.asarray(cols * cut_rat).reshape(-1)[0])
    cr, cc = np.random.randint(rows), np.random.randint(cols)
    r1, c1 = int(np.clip(cr - cut_r // 2, 0, rows)), int(np.clip(cc - cut_c // 2, 0, cols))
    r2, c2 = int(np.clip(cr + cut_r // 2, 0, rows)), int(np.clip(cc + cut_c // 2, 0, cols))
    lam = 1 - ((r2 - r1) * (c2 - c1) / (cols * rows))
    return batch_idx, (r1, c1, r2, c2), float(lam)


def icarl_video_mix_training_step(current_model: nn.Module, batch_data: Dict[str, torch.Tensor], num_classes: int,
                                  video_mix_prob: float, video_mix_alpha, current_task: int = 0,
                                  prev_model: Optional[nn.Module] = None, previous_task_num_classes: int = 0) -> torch.Tensor:
    """``ICARLVideoMix.training_step`` (libs/cil/icarl_video_mix.py:20-45): with probability ``video_mix_prob`` a box of
    every frame is overwritten, in place, by the same box of a permuted sample and the one-hot targets are mixed by the
    box area; then the iCaRL step on the mixed clips.  ``imgs`` must be the (B, T, 3, H, W) tensor (the box copy is a
    strided tensor copy).  The config's ``video_mix_alpha`` reaches ``np.random.beta`` as a 1-tuple in the reference
    (trailing comma at :22); pass the config value, the tuple is made here."""
    imgs, labels = batch_data['imgs'], batch_data['label']
    if not torch.is_tensor(imgs) or imgs.dim() != 5:
        raise TypeError('icarl_video_mix_training_step needs imgs as a (B, T, 3, H, W) tensor')
    alpha = (video_mix_alpha,)
    draw = tubemix_draw(imgs.size(0), imgs.size(-2), imgs.size(-1), alpha, video_mix_prob)
    base = None
    if draw is not None:
        batch_idx, (r1, c1, r2, c2), lam = draw
        perm = batch_idx.to(imgs.device)
        imgs[:, :, :, r1:r2, c1:c2] = imgs[perm][:, :, :, r1:r2, c1:c2]
        flat = labels.reshape(-1).contiguous()
        lam_t = torch.full((flat.numel(),), lam, dtype=torch.float32, device=imgs.device)
        # y * lam + y[batch_idx] * (1 - lam) on one-hot rows = the smooth-label kernel with exponent 1
        base = K.acm_targets(flat, flat[perm].contiguous(), lam_t, 1.0, num_classes)
    cls_score = current_model(imgs, return_loss=False)
    prev_logits = None
    if current_task > 0 and prev_model is not None:
        with torch.no_grad():
            prev_logits = prev_model(imgs, return_loss=False)
    return _soft_ce(cls_score, labels.view(-1), prev_logits=prev_logits, prev_num_classes=previous_task_num_classes,
                    base_targets=base)


import os as _os
_MAIN_HIGH_PRIORITY = _os.environ.get('BDVCIL_MAIN_HIGH_PRIORITY', '0') != '0'


class TrainEngine:
    """One optimisation step = forward, backward (bucketed gradient all-reduce overlapped), clip, fused SGD."""

    def __init__(self, model: nn.Module, optimizer, grad_clip: Optional[float] = None, reducer: Optional[GradAllReducer] = None):
        self.model, self.optimizer, self.grad_clip, self.reducer = model, optimizer, grad_clip, reducer
        if reducer is not None:
            optimizer.set_grad_scale(reducer.grad_scale)

    def step(self, batch_data: Dict[str, torch.Tensor], loss_fn=None) -> Dict[str, torch.Tensor]:
        if _MAIN_HIGH_PRIORITY and torch.cuda.is_available():
            # The dependent chain (forward, BatchNorm backward -> dgrad -> ...) on a high-priority stream; the weight
            # gradients, which nothing waits for until the optimizer, stay on the normal-priority side stream and fill in.
            if getattr(self, '_hi', None) is None:
                self._hi = torch.cuda.Stream(priority=-1)
            cur = torch.cuda.current_stream()
            self._hi.wait_stream(cur)
            with torch.cuda.stream(self._hi):
                losses = self._step(batch_data, loss_fn)
            cur.wait_stream(self._hi)
            return losses
        return self._step(batch_data, loss_fn)

    def _step(self, batch_data: Dict[str, torch.Tensor], loss_fn=None) -> Dict[str, torch.Tensor]:
        self.optimizer.zero_grad(set_to_none=True)
        if loss_fn is None:
            losses = base_training_step(self.model, batch_data)
        else:
            losses = loss_fn(self.model, batch_data)
        losses['loss'].backward()
        if self.reducer is not None:
            self.reducer.finish()
        if self.grad_clip:
            self.optimizer.clip_grad_norm_(self.grad_clip)
        self.optimizer.step()
        return losses
