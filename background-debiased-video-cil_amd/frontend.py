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
