"""CPU restatement of the resampling in the reference's frame pipeline.  TEST INFRASTRUCTURE: only tests/ (and smoke / cpu_baseline)
import this.

Reference call sites: ``Resize(scale=(-1, 256))`` and ``MultiScaleCrop(...)`` + ``Resize(scale=(224, 224), keep_ratio=False)``
(configs/ucf101/bgmix_plus_randAug/bgmix_seed_1000_inc_10_stages_bgmix_plus_randAug.py:127-136); UPSTREAM mmaction2 0.24
``Resize`` -> ``mmcv.imresize(img, (w, h), interpolation='bilinear')`` -> ``cv2.resize(..., interpolation=cv2.INTER_LINEAR)``,
``mmcv.rescale_size`` for the ``(-1, 256)`` form, mmaction2's ``MultiScaleCrop`` for the crop choice.  The algorithm lives in
OpenCV (opencv-python 4.x, modules/imgproc/src/resize.cpp), absent from /root/reference and from this image, as are mmcv / mmaction2:

    **PARITY UNPINNED** -- no golden vector of the reference covers this and nothing here can produce one.  What follows restates
    OpenCV's published 8-bit INTER_LINEAR arithmetic: per axis ``f = float((d + 0.5) * scale - 0.5)``, ``s = floor(f)``, weights
    ``saturate_cast<short>((1 - f) * 2048)`` / ``(f * 2048)`` (round half to even); along x taps outside the image collapse onto the
    edge with weight 2048, along y only the row indices are clipped; rows are combined as
    ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``; an exact 2x shrink uses the 2x2 box mean instead.
    The tests hold the HIP kernel to THIS restatement bit for bit, and this restatement to properties any correct bilinear resize
    has (identity, constants, within one grey level of the real-valued bilinear interpolation)."""
from __future__ import annotations

import random

import numpy as np


def _axis(dsize: int, ssize: int, clamp_fraction: bool):
    scale = 1.0 / (float(dsize) / float(ssize))                         # double, as cv::resize forms it
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:
        lo, hi = s < 0, s >= ssize - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, ssize - 1, s))
    a0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)     # np.rint: half to even
    a1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    s0, s1 = np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1)
    return s0, s1, a0, a1


def resize_linear_u8(img: np.ndarray, Wd: int, Hd: int) -> np.ndarray:
    """``cv2.resize(img, (Wd, Hd), interpolation=cv2.INTER_LINEAR)`` for an (H, W, C) uint8 image."""
    Hs, Ws = img.shape[:2]
    if (Hs, Ws) == (Hd, Wd):
        return img.copy()
    a = img.astype(np.int64)
    if Ws == 2 * Wd and Hs == 2 * Hd:
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xs0, xs1, xa0, xa1 = _axis(Wd, Ws, True)
    ys0, ys1, ya0, ya1 = _axis(Hd, Hs, False)
    rows = a[:, xs0] * xa0[None, :, None] + a[:, xs1] * xa1[None, :, None]                            # horizontal pass, every source row
    out = (((ya0[:, None, None] * (rows[ys0] >> 4)) >> 16) + ((ya1[:, None, None] * (rows[ys1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def rescale_size(w: int, h: int, scale) -> tuple:
    """``mmcv.rescale_size((w, h), scale)`` for the ``(-1, S)`` / ``(a, b)`` forms: the factor that fits the long edge into
    max(scale) and the short edge into min(scale) (``(-1, 256)`` -> np.inf for the long edge: short edge to 256); sizes rounded
    as ``int(x * factor + 0.5)``."""
    a, b = scale
    if a == -1 or b == -1:
        long_edge, short_edge = float('inf'), max(a, b)
    else:
        long_edge, short_edge = max(a, b), min(a, b)
    factor = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def multi_scale_crop_box(img_w: int, img_h: int, input_size=(224, 224), scales=(1, 0.875, 0.75, 0.66), max_wh_scale_gap=1,
                         random_crop=False, num_fixed_crops=13, rng=random) -> tuple:
    """``MultiScaleCrop``'s draw: (x_offset, y_offset, crop_w, crop_h); two ``random.choice`` calls (or one + two ``randint``)."""
    base = min(img_w, img_h)
    sizes = [int(base * s) for s in scales]
    cand = [[w, h] for i, h in enumerate(sizes) for j, w in enumerate(sizes) if abs(i - j) <= max_wh_scale_gap]
    crop = list(rng.choice(cand))
    for i in range(2):
        if abs(crop[i] - input_size[i]) < 3:
            crop[i] = input_size[i]
    cw, ch = crop
    if random_crop:
        return rng.randint(0, img_w - cw), rng.randint(0, img_h - ch), cw, ch
    ws, hs = (img_w - cw) // 4, (img_h - ch) // 4
    offs = [(0, 0), (4 * ws, 0), (0, 4 * hs), (4 * ws, 4 * hs), (2 * ws, 2 * hs)]
    if num_fixed_crops == 13:
        offs += [(0, 2 * hs), (4 * ws, 2 * hs), (2 * ws, 4 * hs), (2 * ws, 0), (ws, hs), (3 * ws, hs), (ws, 3 * hs), (3 * ws, 3 * hs)]
    x, y = rng.choice(offs)
    return x, y, cw, ch


def sample_frames(total_frames: int, num_clips: int = 8, clip_len: int = 1, frame_interval: int = 1, test_mode: bool = False,
                  start_index: int = 1, rng=np.random) -> np.ndarray:
    """UPSTREAM mmaction2 0.24 ``SampleFrames`` (no temporal jitter, out_of_bound_opt='loop', twice_sample=False): 1-based frame
    numbers of the ``img_{:05}.jpg`` files, shape (num_clips * clip_len,)."""
    ori = clip_len * frame_interval
    if test_mode:
        avg = (total_frames - ori + 1) / float(num_clips)
        if total_frames > ori - 1:
            offsets = (np.arange(num_clips) * avg + avg / 2.0).astype(np.int64)
        else:
            offsets = np.zeros((num_clips,), dtype=np.int64)
    else:
        avg = (total_frames - ori + 1) // num_clips
        if avg > 0:
            offsets = np.arange(num_clips) * avg + rng.randint(avg, size=num_clips)
        elif total_frames > max(num_clips, ori):
            offsets = np.sort(rng.randint(total_frames - ori + 1, size=num_clips))
        elif avg == 0:
            ratio = (total_frames - ori + 1.0) / num_clips
            offsets = np.around(np.arange(num_clips) * ratio)
        else:
            offsets = np.zeros((num_clips,), dtype=np.int64)
    inds = offsets[:, None] + np.arange(clip_len)[None, :] * frame_interval
    inds = np.concatenate(inds).reshape((-1, clip_len))
    inds = np.mod(inds, total_frames)
    return np.concatenate(inds).astype(np.int64) + start_index
