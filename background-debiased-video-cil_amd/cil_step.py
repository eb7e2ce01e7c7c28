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
                raise NotImplementedError('kd_exemplar_only=True is not used by the shipped configs (cil.py:529-536)')
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
    """libs/cil/icarl.py:97-125 (without the ActorCutMix ``foreground_ratio`` branch, which is out of scope).
    The prev model is evaluated on the whole batch and only old-class rows are used (same values as the
    reference's ``imgs[indices]`` gather: eval-mode BN makes rows independent)."""
    imgs, targets = batch_data['imgs'], batch_data['label']
    cls_score = current_model(imgs, return_loss=False)
    prev_logits = None
    if current_task > 0 and prev_model is not None:
        with torch.no_grad():
            prev_logits = prev_model(imgs, return_loss=False)
    return _soft_ce(cls_score, targets.view(-1), prev_logits=prev_logits, prev_num_classes=previous_task_num_classes)


class TrainEngine:
    """One optimisation step = forward, backward (bucketed gradient all-reduce overlapped), clip, fused SGD."""

    def __init__(self, model: nn.Module, optimizer, grad_clip: Optional[float] = None, reducer: Optional[GradAllReducer] = None):
        self.model, self.optimizer, self.grad_clip, self.reducer = model, optimizer, grad_clip, reducer
        if reducer is not None:
            optimizer.set_grad_scale(reducer.grad_scale)

    def step(self, batch_data: Dict[str, torch.Tensor], loss_fn=None) -> Dict[str, torch.Tensor]:
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
