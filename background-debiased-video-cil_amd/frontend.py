"""Fused background-mix + normalize front-end (device side of BackgroundMixDataset).

Arithmetic of libs/loader/comix_loader.py:72-75 (bg: Normalize(bg_mean, bg_std)), :138-145 (blend on normalised
tensors) and UPSTREAM Normalize with ``img_norm_cfg`` (configs/.../bgmix_seed_1000_...:121-122), for a whole batch in
one HBM pass: uint8 frames + uint8 background in, NHWC4 fp32 out (the stem's input layout).  The mix decision
(``not randAug`` or ``random() < prob``, comix_loader.py:110-116) stays with the caller and arrives as ``mix``.
File discovery / decoding / resize / crop are out of scope.
"""
from __future__ import annotations

import torch

from . import kernels as K
from .resnet_tsm import Nhwc4Frames

IMG_MEAN = (123.675, 116.28, 103.53)
IMG_STD = (58.395, 57.12, 57.375)


class BackgroundMixFrontEnd:
    def __init__(self, alpha: float = 0.5, mean=IMG_MEAN, std=IMG_STD):
        self.alpha, self.mean, self.std = float(alpha), tuple(mean), tuple(std)

    def __call__(self, frames_u8: torch.Tensor, bg_u8: torch.Tensor = None, mix: torch.Tensor = None) -> Nhwc4Frames:
        """frames_u8 (B,T,H,W,3), bg_u8 (B,H,W,3), mix (B,) bool -> Nhwc4Frames (B*T,H,W,4)."""
        o4, _ = K.bgmix_normalize_u8(frames_u8, bg_u8, mix, self.alpha, self.mean, self.std, True, False)
        return Nhwc4Frames(o4, frames_u8.shape[0], frames_u8.shape[1])

    def as_nchw(self, frames_u8, bg_u8=None, mix=None) -> torch.Tensor:
        """Same arithmetic, output shaped like the reference's collated batch: (B,T,3,H,W) fp32."""
        _, oc = K.bgmix_normalize_u8(frames_u8, bg_u8, mix, self.alpha, self.mean, self.std, False, True)
        return oc


class TrainClipFrontEnd:
    """Device side of the train pipeline for a whole batch: RandAugment on the uint8 frames, then the
    ``BackgroundMixDataset`` decision and blend.

    Decision rule of ``prepare_train_frames`` (comix_loader.py:105-116): with ``with_randAug`` a sample is mixed with a
    background exactly when RandAugment did not fire for it; otherwise with probability ``prob`` (one
    ``random.random()`` per sample).  ``crop_resize`` is the hook for the stages the reference runs between the two
    (MultiScaleCrop + Resize, configs/...bgmix_plus_randAug.py:131-138), applied to the uint8 clips when given.
    Returns ``(Nhwc4Frames, randAug (B,) bool, mixed (B,) bool)``; ``randAug`` is what the reference collects into the
    batch under that key."""

    def __init__(self, randaug=None, alpha: float = 0.5, prob: float = 0.25, with_randAug: bool = True, crop_resize=None):
        self.randaug, self.prob, self.with_randAug, self.crop_resize = randaug, prob, with_randAug, crop_resize
        self.mix = BackgroundMixFrontEnd(alpha=alpha)

    def decide(self, frames_u8: torch.Tensor):
        B = frames_u8.shape[0]
        if self.with_randAug:
            if self.randaug is None:
                raise ValueError('with_randAug=True needs a RandAugment stage (comix_loader.py:109-111 reads result["randAug"])')
            frames_u8, rand_flags = self.randaug(frames_u8)
            return frames_u8, rand_flags, ~rand_flags
        import random
        rand_flags = torch.zeros(B, dtype=torch.bool, device=frames_u8.device)
        if self.randaug is not None:
            frames_u8, rand_flags = self.randaug(frames_u8)
        mixed = torch.tensor([random.random() < self.prob for _ in range(B)], dtype=torch.bool, device=frames_u8.device)
        return frames_u8, rand_flags, mixed

    def __call__(self, frames_u8: torch.Tensor, bg_u8: torch.Tensor, as_nchw: bool = False):
        frames_u8, rand_flags, mixed = self.decide(frames_u8)
        if self.crop_resize is not None:
            frames_u8, bg_u8 = self.crop_resize(frames_u8, bg_u8)
        out = self.mix.as_nchw(frames_u8, bg_u8, mixed) if as_nchw else self.mix(frames_u8, bg_u8, mixed)
        return out, rand_flags, mixed
