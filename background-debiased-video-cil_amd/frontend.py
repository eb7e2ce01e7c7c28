"""Fused background-mix + normalize front-end (device side of BackgroundMixDataset).

Arithmetic of libs/loader/comix_loader.py:72-75 (bg: Normalize(bg_mean, bg_std)), :138-145 (blend on normalised
tensors) and UPSTREAM Normalize with ``img_norm_cfg`` (configs/.../bgmix_seed_1000_...:121-122), for a whole batch in
one HBM pass: uint8 frames + uint8 background in, NHWC4 fp32 out (the stem's input layout).  The mix decision
(``not randAug`` or ``random() < prob``, comix_loader.py:110-116) stays with the caller and arrives as ``mix``.
The background's ``Resize -> RandomCrop`` (comix_loader.py:72-73) is ``BackgroundCropFrontEnd``; file discovery and JPEG decoding
are out of scope.
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
        """frames_u8 (B,T,H,W,3), bg_u8 (B,H,W,3) uint8 (or the fp32 output of ``BackgroundCropFrontEnd``), mix (B,) bool -> Nhwc4Frames
        (B*T,H,W,4)."""
        o4, _ = K.bgmix_normalize_u8(frames_u8, bg_u8, mix, self.alpha, self.mean, self.std, True, False)
        return Nhwc4Frames(o4, frames_u8.shape[0], frames_u8.shape[1])

    def as_nchw(self, frames_u8, bg_u8=None, mix=None) -> torch.Tensor:
        """Same arithmetic, output shaped like the reference's collated batch: (B,T,3,H,W) fp32."""
        _, oc = K.bgmix_normalize_u8(frames_u8, bg_u8, mix, self.alpha, self.mean, self.std, False, True)
        return oc


class BackgroundCropFrontEnd:
    """``Resize(bg_resize)`` -> ``RandomCrop(bg_crop_size)`` of ``BackgroundMixDataset.bg_pipeline`` (libs/loader/comix_loader.py:72-73;
    the constructor defaults ``bg_resize=256``, ``bg_crop_size=(224, 224)`` of :27-28) for a batch of uint8 background images of one size, in one kernel; its fp32 output
    goes straight into ``BackgroundMixFrontEnd`` (which applies the pipeline's ``Normalize`` and the blend).  The crop offsets are
    drawn as torchvision's ``RandomCrop.get_params`` draws them -- per image ``torch.randint(0, h - th + 1, (1,))`` then
    ``torch.randint(0, w - tw + 1, (1,))`` on torch's global CPU generator, and no draw at all when the resized image already has
    the crop's size -- so a seeded run picks the reference's crops.  Parity unpinned: torchvision is not importable here (its
    version is not pinned by the reference either); the resampling follows ATen's bilinear kernel, which torchvision calls."""

    def __init__(self, resize: int = 256, crop_size=(224, 224)):
        self.resize = int(resize)
        self.crop = (int(crop_size), int(crop_size)) if isinstance(crop_size, int) else (int(crop_size[0]), int(crop_size[1]))   # (h, w)

    def draw(self, batch: int, h: int, w: int):
        th, tw = self.crop
        if h < th or w < tw:
            raise ValueError(f'Required crop size {(th, tw)} is larger than input image size {(h, w)}')
        tops, lefts = [], []
        for _ in range(batch):
            if (h, w) == (th, tw):
                i = j = 0
            else:
                i = int(torch.randint(0, h - th + 1, size=(1,)).item())
                j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
            tops.append(i)
            lefts.append(j)
        return tops, lefts

    def __call__(self, bg_u8: torch.Tensor, offsets=None) -> torch.Tensor:
        """bg_u8 (B,Hs,Ws,3) uint8 -> (B,crop,crop,3) fp32 pixel values.  ``offsets = (tops, lefts)`` overrides the draws."""
        B, Hs, Ws, _ = bg_u8.shape
        Hr, Wr = K.resized_size(Hs, Ws, self.resize)
        tops, lefts = self.draw(B, Hr, Wr) if offsets is None else offsets
        dev = bg_u8.device
        return K.bg_resize_crop_u8(bg_u8, self.resize, self.crop[0], self.crop[1],
                                   torch.tensor(list(tops), dtype=torch.int32).to(dev), torch.tensor(list(lefts), dtype=torch.int32).to(dev))


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


def crop_offsets(kind: str, img_h: int, img_w: int, crop_h: int, crop_w: int):
    """``(x_offset, y_offset, flip)`` of the fixed test-time crop transforms, in emission order.

    ``FiveCrop``: the reference's own transform (libs/pipelines/five_crops.py:81-92): corners on a quarter-step grid, then
    the centre.  ``TenCrop`` (UPSTREAM, the class FiveCrop was derived from; :95,:98 keep its flip lines as comments): every
    crop is followed by its horizontal flip.  ``ThreeCrop`` / ``CenterCrop``: UPSTREAM mmaction2 0.x semantics."""
    if crop_h > img_h or crop_w > img_w:
        raise ValueError(f'crop {crop_h}x{crop_w} larger than the {img_h}x{img_w} frame')
    if kind in ('FiveCrop', 'TenCrop'):
        w_step, h_step = (img_w - crop_w) // 4, (img_h - crop_h) // 4
        offs = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
        if kind == 'FiveCrop':
            return [(x, y, 0) for x, y in offs]
        return [(x, y, f) for x, y in offs for f in (0, 1)]
    if kind == 'ThreeCrop':
        if crop_h == img_h:
            w_step = (img_w - crop_w) // 2
            return [(0, 0, 0), (2 * w_step, 0, 0), (w_step, 0, 0)]
        if crop_w == img_w:
            h_step = (img_h - crop_h) // 2
            return [(0, 0, 0), (0, 2 * h_step, 0), (0, h_step, 0)]
        raise ValueError('ThreeCrop needs the crop to span the full height or width of the frame')
    if kind == 'CenterCrop':
        return [((img_w - crop_w) // 2, (img_h - crop_h) // 2, 0)]
    raise KeyError(f'unknown crop transform {kind!r}')


class CropFrontEnd:
    """Device side of the val / test pipelines after decode + Resize: ``CenterCrop`` / ``ThreeCrop`` / ``FiveCrop`` /
    ``TenCrop`` and ``Normalize`` in one pass.  uint8 frames (B,T,H,W,3) -> the (B, crops*T, ...) clip batch that
    ``FormatShape('NCHW')`` + collate produce (crop-major), as ``Nhwc4Frames`` for the stem or as an NCHW tensor."""

    def __init__(self, kind: str = 'TenCrop', crop_size=256, mean=IMG_MEAN, std=IMG_STD):
        self.kind = kind
        self.crop_w, self.crop_h = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)   # (w, h) as mmaction
        self.mean, self.std = tuple(mean), tuple(std)
        crop_offsets(kind, self.crop_h, self.crop_w, self.crop_h, self.crop_w)      # validates `kind` early

    def _run(self, frames_u8, nhwc4: bool):
        B, T, H, W, _ = frames_u8.shape
        crops = crop_offsets(self.kind, H, W, self.crop_h, self.crop_w)
        o4, oc = K.crop_normalize_u8(frames_u8, crops, self.crop_h, self.crop_w, self.mean, self.std, nhwc4, not nhwc4)
        return (o4, oc, B, len(crops) * T)

    def __call__(self, frames_u8: torch.Tensor) -> Nhwc4Frames:
        o4, _, B, n = self._run(frames_u8, True)
        return Nhwc4Frames(o4, B, n)

    def as_nchw(self, frames_u8: torch.Tensor) -> torch.Tensor:
        return self._run(frames_u8, False)[1]


class MultiScaleCropResize:
    """``MultiScaleCrop(input_size, scales, max_wh_scale_gap, random_crop, num_fixed_crops)`` + ``Resize(scale=(w, h), keep_ratio=False)``
    of the train pipeline (configs/ucf101/bgmix_plus_randAug/...py:129-136; UPSTREAM mmaction2 0.24 semantics) as the ``crop_resize``
    stage of ``TrainClipFrontEnd``: one crop box per clip (all frames of a sample share it), drawn with two ``random.choice`` calls
    as upstream does (sizes first, then one of the 5 / 13 fixed offsets; ``random_crop=True``: two ``random.randint``), then crop +
    ``cv2.resize(INTER_LINEAR)`` in one kernel (``bdv_resize_linear_u8``; parity unpinned, see include/bdvcil_hip.h)."""

    def __init__(self, input_size=224, scales=(1, 0.875, 0.75, 0.66), max_wh_scale_gap=1, random_crop=False, num_fixed_crops=5,
                 out_size=None):
        self.input_size = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)       # (w, h)
        self.scales, self.gap, self.random_crop = tuple(scales), int(max_wh_scale_gap), bool(random_crop)
        if num_fixed_crops not in (5, 13):
            raise ValueError(f'num_fixed_crops must be 5 or 13, got {num_fixed_crops}')
        self.num_fixed_crops = num_fixed_crops
        self.out_size = self.input_size if out_size is None else tuple(out_size)                               # (w, h) of the Resize after it

    def draw(self, img_w: int, img_h: int):
        import random
        base = min(img_w, img_h)
        sizes = [int(base * s) for s in self.scales]
        cand = [[w, h] for i, h in enumerate(sizes) for j, w in enumerate(sizes) if abs(i - j) <= self.gap]
        crop = list(random.choice(cand))
        for i in range(2):
            if abs(crop[i] - self.input_size[i]) < 3:
                crop[i] = self.input_size[i]
        cw, ch = crop
        if self.random_crop:
            return random.randint(0, img_w - cw), random.randint(0, img_h - ch), cw, ch
        ws, hs = (img_w - cw) // 4, (img_h - ch) // 4
        offs = [(0, 0), (4 * ws, 0), (0, 4 * hs), (4 * ws, 4 * hs), (2 * ws, 2 * hs)]
        if self.num_fixed_crops == 13:
            offs += [(0, 2 * hs), (4 * ws, 2 * hs), (2 * ws, 4 * hs), (2 * ws, 0), (ws, hs), (3 * ws, hs), (ws, 3 * hs), (3 * ws, 3 * hs)]
        x, y = random.choice(offs)
        return x, y, cw, ch

    def __call__(self, frames_u8: torch.Tensor, bg=None, boxes=None):
        B, _, H, W, _ = frames_u8.shape
        if boxes is None:
            boxes = [self.draw(W, H) for _ in range(B)]
        self.last_boxes = boxes
        return K.resize_linear_u8(frames_u8, self.out_size[1], self.out_size[0], boxes), bg
