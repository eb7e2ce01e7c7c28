"""RandAugment on the GPU for whole batches of uint8 clips.

Mirrors ``libs/pipelines/rand_augment.py``: the operation table of ``augment_list()`` (:163-220), the magnitude rule and
the per-clip random draws of ``RandAugment.__call__`` / ``_rand_aug`` (:223-264) -- in the same order from the same
generators (``random`` and ``np.random``), so a seeded run makes the reference's decisions -- and turns each drawn
operation into one row of the device tables ``kernels.randaug_apply`` consumes.  All pixel work is in
``csrc/augment.hip`` and is bit-identical to the Pillow calls of the reference; the small amount of double-precision
parameter arithmetic the reference does in Python (rotation matrix, cut-out rectangle, 16.16 conversion) is done here
in Python too, with the same expressions.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import kernels as K

FILL_COLOR = (124, 116, 104)                      # rand_augment.py:15

# operation codes of bdv_randaug_apply (include/bdvcil_hip.h)
IDENTITY, AUTOCONTRAST, EQUALIZE, SOLARIZE, POSTERIZE, COLOR, CONTRAST, BRIGHTNESS, SHARPNESS, AFFINE_FIXED, AFFINE_SCALE, CUTOUT = range(12)


def augment_list() -> List[Tuple[str, float, float]]:
    """(name, minval, maxval) in the reference's order (rand_augment.py:201-218, the FixMatch table)."""
    return [('Identity', 0., 1.0), ('AutoContrast', 0, 1), ('Equalize', 0, 1), ('Rotate', 0, 30), ('Solarize', 0, 256),
            ('Color', 0.05, 0.95), ('Contrast', 0.05, 0.95), ('Brightness', 0.05, 0.95), ('Sharpness', 0.05, 0.95),
            ('ShearX', 0., 0.3), ('TranslateX', 0., 0.3), ('TranslateY', 0., 0.3), ('Posterize', 4, 8), ('ShearY', 0., 0.3),
            ('CutoutAbs', 0, 112)]


def _fix(v: float) -> int:
    """Pillow Geometry.c: FIX(v) = FLOOR(v * 65536.0 + 0.5), wrapped to a C int."""
    x = v * 65536.0 + 0.5
    r = int(x) if x >= 0.0 else int(math.floor(x))
    r &= 0xFFFFFFFF
    return r - (1 << 32) if r & 0x80000000 else r


def _affine_row(a: Sequence[float], fill: int):
    """Row for ``img.transform(size, AFFINE, a, fillcolor=...)`` with nearest resampling (ImagingTransformAffine)."""
    a = [float(v) for v in a]
    if a[1] == 0 and a[3] == 0:                   # no cross terms: ImagingScaleAffine (translations)
        return [AFFINE_SCALE, 0, 0, 0, 0, 0, 0, fill], [a[0], a[2], a[4], a[5]]
    fixed = [_fix(a[0]), _fix(a[1]), _fix(a[2] + a[0] * 0.5 + a[1] * 0.5), _fix(a[3]), _fix(a[4]), _fix(a[5] + a[3] * 0.5 + a[4] * 0.5)]
    return [AFFINE_FIXED] + fixed + [fill], [0.0] * 4


def _rotate_matrix(angle: float, w: int, h: int):
    """PIL.Image.Image.rotate (expand=False, centre = image centre): the matrix handed to transform(AFFINE)."""
    cx, cy = w / 2.0, h / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def op_row(name: str, val: float, flip_sign: bool, init_loc: Tuple[float, float], H: int, W: int,
           fillcolor: Tuple[int, int, int] = FILL_COLOR):
    """One table row (8 ints, 4 doubles) for operation ``name`` at magnitude ``val`` (rand_augment.py:17-160)."""
    fill = (fillcolor[0] << 16) | (fillcolor[1] << 8) | fillcolor[2]
    zero_i, zero_d = [0] * 7, [0.0] * 4
    if name in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
        lim = {'ShearX': 0.3, 'ShearY': 0.3, 'TranslateX': 0.45, 'TranslateY': 0.45, 'Rotate': 30}[name]
        assert -lim <= val <= lim
        v = -val if flip_sign else val
        if name == 'ShearX':
            return _affine_row((1, v, 0, 0, 1, 0), fill)
        if name == 'ShearY':
            return _affine_row((1, 0, 0, v, 1, 0), fill)
        if name == 'TranslateX':
            return _affine_row((1, 0, v * W, 0, 1, 0), fill)
        if name == 'TranslateY':
            return _affine_row((1, 0, 0, 0, 1, v * H), fill)
        angle = v % 360.0
        if angle == 0:                            # Image.rotate returns a copy
            return [IDENTITY] + zero_i, zero_d
        if angle in (90.0, 180.0, 270.0):
            raise NotImplementedError('Rotate by a multiple of 90 degrees takes Pillow\'s transpose path (not reachable: |v| <= 30)')
        return _affine_row(_rotate_matrix(angle, W, H), fill)
    if name == 'CutoutAbs':
        if val < 0:
            return [IDENTITY] + zero_i, zero_d
        x0, y0 = init_loc
        x0 = int(max(0, x0 - val / 2.))
        y0 = int(max(0, y0 - val / 2.))
        x1 = min(W, x0 + val)
        y1 = min(H, y0 + val)
        return [CUTOUT, int(x0), int(y0), int(x1), int(y1), 0, 0, fill], zero_d     # ImageDraw: (int) casts, inclusive
    if name == 'Identity':
        return [IDENTITY] + zero_i, zero_d
    if name == 'AutoContrast':
        return [AUTOCONTRAST] + zero_i, zero_d
    if name == 'Equalize':
        return [EQUALIZE] + zero_i, zero_d
    if name == 'Solarize':
        assert 0 <= val <= 256
        return [SOLARIZE] + zero_i, [float(val), 0.0, 0.0, 0.0]
    if name == 'Posterize':
        return [POSTERIZE, max(1, int(val))] + [0] * 6, zero_d
    if name in ('Color', 'Contrast', 'Brightness', 'Sharpness'):
        assert 0.05 <= val <= 1.9
        if val > 1.0:
            raise NotImplementedError('enhancement factors above 1 take Pillow\'s clipping blend (not reachable: maxval 0.95)')
        code = {'Color': COLOR, 'Contrast': CONTRAST, 'Brightness': BRIGHTNESS, 'Sharpness': SHARPNESS}[name]
        return [code] + zero_i, [float(val), 0.0, 0.0, 0.0]
    raise KeyError(f'unknown RandAugment operation {name!r}')


class RandAugment:
    """``RandAugment(n, m, prob)`` of the reference for a batch: ``__call__(frames_u8 (B,T,H,W,3) cuda uint8)`` returns
    ``(augmented frames, randAug (B,) bool)``.  Sample b draws exactly what the reference's b-th ``__call__`` would."""

    def __init__(self, n: int, m: int, prob: float = 0.5):
        self.n, self.m, self.prob = n, m, prob
        self.augment_list = augment_list()

    def draw(self, H: int, W: int):
        """The random decisions of one sample in the reference's order (rand_augment.py:230-245); None = not augmented."""
        if not (random.random() < self.prob):
            return None
        ops = random.choices(self.augment_list, k=self.n)
        flip_sign = random.random() > 0.5
        x0 = np.random.uniform(W)                # sic: low = W, high = 1.0 (rand_augment.py:242-243)
        y0 = np.random.uniform(H)
        return ops, flip_sign, (x0, y0)

    def rows(self, draws, H: int, W: int):
        """Per-slot tables for a list of ``draw`` results: n pairs (op_i (B,8) int32, op_d (B,4) float64) on the CPU."""
        B = len(draws)
        out = []
        for slot in range(self.n):
            oi = np.zeros((B, 8), np.int32)
            od = np.zeros((B, 4), np.float64)
            for b, d in enumerate(draws):
                if d is None:
                    continue
                (name, minval, maxval), flip_sign, init_loc = d[0][slot], d[1], d[2]
                val = (float(self.m) / 30) * float(maxval - minval) + minval
                ri, rd = op_row(name, val, flip_sign, init_loc, H, W)
                oi[b], od[b] = np.array(ri, np.int64).astype(np.int32), rd
            out.append((torch.from_numpy(oi), torch.from_numpy(od)))
        return out

    def apply_draws(self, frames_u8: torch.Tensor, draws) -> torch.Tensor:
        B, T, H, W, _ = frames_u8.shape
        cur = frames_u8
        for oi, od in self.rows(draws, H, W):
            if not bool((oi[:, 0] != IDENTITY).any()):
                continue
            cur = K.randaug_apply(cur, oi.to(frames_u8.device, non_blocking=True), od.to(frames_u8.device, non_blocking=True))
        return cur

    def __call__(self, frames_u8: torch.Tensor):
        B, T, H, W, _ = frames_u8.shape
        draws = [self.draw(H, W) for _ in range(B)]
        flags = torch.tensor([d is not None for d in draws], dtype=torch.bool, device=frames_u8.device)
        return self.apply_draws(frames_u8, draws), flags
