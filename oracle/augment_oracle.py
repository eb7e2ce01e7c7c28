"""CPU oracle for the RandAugment stage of the data path (TEST INFRASTRUCTURE ONLY: imported by tests/ and never by the
product package).

Restates, in numpy, what ``libs/pipelines/rand_augment.py`` computes through Pillow for the fifteen operations of
``augment_list()`` (:163-220) and the per-clip draws of ``RandAugment.__call__`` / ``_rand_aug`` (:223-264).  The
arithmetic lives in a third-party dependency (Pillow; 12.2.0 in this image, not vendored by the reference): each function
below restates the corresponding Pillow routine -- named in its docstring -- and is pinned two ways: against golden
vectors produced by the reference's own file (tests/golden/make_golden_randaug.py, bit-exact) and, where Pillow is
importable, against Pillow itself on fresh random frames (tests/test_augment_cpu.py).
Frames are (H, W, 3) uint8 RGB arrays.
"""
import math
import random

import numpy as np

FILL_COLOR = (124, 116, 104)                                      # rand_augment.py:15


def _lut_apply(img, lut):
    out = np.empty_like(img)
    for c in range(3):
        out[..., c] = lut[c][img[..., c]]
    return out


def _hist(img):
    return np.stack([np.bincount(img[..., c].ravel(), minlength=256) for c in range(3)])


def autocontrast(img):
    """ImageOps.autocontrast(image, cutoff=0) (rand_augment.py:71-72)."""
    h = _hist(img)
    lut = np.zeros((3, 256), np.uint8)
    for c in range(3):
        nz = np.nonzero(h[c])[0]
        lo, hi = int(nz[0]), int(nz[-1])
        if hi <= lo:
            lut[c] = np.arange(256)
            continue
        scale = 255.0 / (hi - lo)
        offset = -lo * scale
        for ix in range(256):
            lut[c, ix] = min(max(int(ix * scale + offset), 0), 255)
    return _lut_apply(img, lut)


def equalize(img):
    """ImageOps.equalize(image) (rand_augment.py:79-80); Image.point clips table entries to 255."""
    h = _hist(img)
    lut = np.zeros((3, 256), np.int64)
    for c in range(3):
        occupied = h[c][h[c] > 0]
        step = (int(occupied.sum()) - int(occupied[-1])) // 255 if len(occupied) > 1 else 0
        if not step:
            lut[c] = np.arange(256)
            continue
        n = step // 2
        for i in range(256):
            lut[c, i] = n // step
            n += int(h[c][i])
    return _lut_apply(img, np.clip(lut, 0, 255).astype(np.uint8))


def solarize(img, threshold):
    """ImageOps.solarize (rand_augment.py:87-89)."""
    i = np.arange(256)
    table = np.where(i < threshold, i, 255 - i).astype(np.uint8)
    return _lut_apply(img, np.stack([table] * 3))


def posterize(img, v):
    """ImageOps.posterize with bits = max(1, int(v)) (rand_augment.py:101-104)."""
    bits = max(1, int(v))
    table = (np.arange(256) & ~(2 ** (8 - bits) - 1)).astype(np.uint8)
    return _lut_apply(img, np.stack([table] * 3))


def _blend(deg, img, alpha):
    """Image.blend(deg, img, alpha), 0 <= alpha <= 1: (UINT8)(in1 + alpha * (in2 - in1)) in single precision (Blend.c)."""
    a = np.float32(alpha)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    t = (a * (i - d).astype(np.float32)).astype(np.float32)
    return (d.astype(np.float32) + t).astype(np.float32).astype(np.uint8)


def _gray(img):
    """Image.convert('L'): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16 (Convert.c, L24)."""
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def color(img, v):
    """ImageEnhance.Color(img).enhance(v) (rand_augment.py:112-114)."""
    return _blend(np.stack([_gray(img)] * 3, -1), img, v)


def contrast(img, v):
    """ImageEnhance.Contrast(img).enhance(v): degenerate = int(mean of L + 0.5) everywhere (rand_augment.py:107-109)."""
    g = _gray(img)
    hh = np.bincount(g.ravel(), minlength=256)
    s = 0.0
    for j in range(256):
        s += j * int(hh[j])
    return _blend(np.full_like(img, int(s / g.size + 0.5)), img, v)


def brightness(img, v):
    """ImageEnhance.Brightness(img).enhance(v) (rand_augment.py:117-119)."""
    return _blend(np.zeros_like(img), img, v)


def _smooth(img):
    """image.filter(ImageFilter.SMOOTH): ImagingFilter3x3 with (1,1,1,1,5,1,1,1,1)/13 as floats, offset 0.5 for rounding,
    rows accumulated bottom row first, each row product summed left to right, edges copied (Filter.c)."""
    k = (np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], np.float32) / np.float32(13)).astype(np.float32)
    H, W, _ = img.shape
    out = img.copy()
    f = img.astype(np.float32)

    def row_dot(rows, kk):
        return ((rows[:, :-2] * kk[0]).astype(np.float32) + (rows[:, 1:-1] * kk[1]).astype(np.float32)).astype(np.float32) \
            + (rows[:, 2:] * kk[2]).astype(np.float32)
    ss = np.full((H - 2, W - 2, 3), np.float32(0.5), np.float32)
    ss = (ss + row_dot(f[2:], k[0:3])).astype(np.float32)
    ss = (ss + row_dot(f[1:-1], k[3:6])).astype(np.float32)
    ss = (ss + row_dot(f[:-2], k[6:9])).astype(np.float32)
    out[1:-1, 1:-1] = np.where(ss <= 0, 0, np.where(ss >= 255, 255, ss.astype(np.int32))).astype(np.uint8)
    return out


def sharpness(img, v):
    """ImageEnhance.Sharpness(img).enhance(v) (rand_augment.py:122-124)."""
    return _blend(_smooth(img), img, v)


def _fix(v):
    x = v * 65536.0 + 0.5
    return int(x) if x >= 0 else int(math.floor(x))


def affine_nearest(img, a, fill=FILL_COLOR):
    """img.transform(img.size, AFFINE, a, fillcolor=fill), nearest (Geometry.c: ImagingScaleAffine when a[1] == a[3] == 0,
    else affine_fixed in 16.16 fixed point)."""
    H, W, _ = img.shape
    out = np.empty_like(img)
    out[:] = np.array(fill, np.uint8)
    a = [float(v) for v in a]
    if a[1] == 0 and a[3] == 0:
        xo, yo = a[2] + a[0] * 0.5, a[5] + a[4] * 0.5
        xin = np.full(W, -1)
        for x in range(W):
            xi = -1 if xo < 0.0 else int(xo)
            if 0 <= xi < W:
                xin[x] = xi
            xo += a[0]
        ok = xin >= 0
        for y in range(H):
            yi = -1 if yo < 0.0 else int(yo)
            if 0 <= yi < H:
                out[y, ok] = img[yi, xin[ok]]
            yo += a[4]
        return out
    a0, a1, a3, a4 = _fix(a[0]), _fix(a[1]), _fix(a[3]), _fix(a[4])
    a2 = _fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = _fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    ys, xs = np.mgrid[0:H, 0:W]
    xx, yy = (a2 + ys * a1 + xs * a0) >> 16, (a5 + ys * a4 + xs * a3) >> 16
    ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
    out[ok] = img[yy[ok], xx[ok]]
    return out


def rotate(img, angle, fill=FILL_COLOR):
    """img.rotate(angle, fillcolor=fill) for angles that are not multiples of 90 (Image.rotate builds the matrix)."""
    H, W, _ = img.shape
    angle = angle % 360.0
    if angle == 0:
        return img.copy()
    cx, cy = W / 2.0, H / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return affine_nearest(img, m, fill)


def cutout_abs(img, v, init_loc, fill=FILL_COLOR):
    """CutoutAbs (rand_augment.py:137-155): ImageDraw.rectangle truncates the corners to int and fills them inclusively."""
    if v < 0:
        return img
    H, W, _ = img.shape
    x0, y0 = init_loc
    x0 = int(max(0, x0 - v / 2.))
    y0 = int(max(0, y0 - v / 2.))
    x1, y1 = int(min(W, x0 + v)), int(min(H, y0 + v))
    out = img.copy()
    out[max(y0, 0):min(y1, H - 1) + 1, max(x0, 0):min(x1, W - 1) + 1] = np.array(fill, np.uint8)
    return out


def apply_op(name, img, val, flip_sign=False, init_loc=(0.0, 0.0)):
    """One operation of the table on one frame, dispatched as in _rand_aug (rand_augment.py:250-263)."""
    H, W, _ = img.shape
    s = -val if flip_sign else val
    if name == 'Identity':
        return img
    if name == 'AutoContrast':
        return autocontrast(img)
    if name == 'Equalize':
        return equalize(img)
    if name == 'Rotate':
        return rotate(img, s)
    if name == 'Solarize':
        return solarize(img, val)
    if name == 'Color':
        return color(img, val)
    if name == 'Contrast':
        return contrast(img, val)
    if name == 'Brightness':
        return brightness(img, val)
    if name == 'Sharpness':
        return sharpness(img, val)
    if name == 'ShearX':
        return affine_nearest(img, (1, s, 0, 0, 1, 0))
    if name == 'ShearY':
        return affine_nearest(img, (1, 0, 0, s, 1, 0))
    if name == 'TranslateX':
        return affine_nearest(img, (1, 0, s * W, 0, 1, 0))
    if name == 'TranslateY':
        return affine_nearest(img, (1, 0, 0, 0, 1, s * H))
    if name == 'Posterize':
        return posterize(img, val)
    if name == 'CutoutAbs':
        return cutout_abs(img, val, init_loc)
    raise KeyError(name)


OP_TABLE = [('Identity', 0., 1.0), ('AutoContrast', 0, 1), ('Equalize', 0, 1), ('Rotate', 0, 30), ('Solarize', 0, 256),
            ('Color', 0.05, 0.95), ('Contrast', 0.05, 0.95), ('Brightness', 0.05, 0.95), ('Sharpness', 0.05, 0.95),
            ('ShearX', 0., 0.3), ('TranslateX', 0., 0.3), ('TranslateY', 0., 0.3), ('Posterize', 4, 8), ('ShearY', 0., 0.3),
            ('CutoutAbs', 0, 112)]


def rand_augment(frames, n, m, prob):
    """RandAugment(n, m, prob).__call__ on one sample's frame list (rand_augment.py:230-264), drawing from ``random`` and
    ``np.random`` in the reference's order.  Returns (frames, randAug flag, names of the drawn operations)."""
    if not (random.random() < prob):
        return frames, False, []
    ops = random.choices(OP_TABLE, k=n)
    flip_sign = random.random() > 0.5
    H, W, _ = frames[0].shape
    x0 = np.random.uniform(W)
    y0 = np.random.uniform(H)
    frames = list(frames)
    for name, minval, maxval in ops:
        val = (float(m) / 30) * float(maxval - minval) + minval
        frames = [apply_op(name, f, val, flip_sign, (x0, y0)) for f in frames]
    return frames, True, [o[0] for o in ops]


# ---- fixed test-time crops ----------------------------------------------------------------------------------------------

def crop_frames(frames, kind, crop_size):
    """FiveCrop as the reference ships it (libs/pipelines/five_crops.py:77-100): quarter-step corner offsets then the
    centre, each crop applied to all frames before the next.  TenCrop = the class it was derived from: the commented
    lines :95 and :98 put the horizontally flipped copies right after each crop.  ThreeCrop / CenterCrop: mmaction2 0.x
    (UPSTREAM, not in the reference tree).  frames: list of (H, W, 3) arrays -> list of (crop_h, crop_w, 3) arrays."""
    crop_w, crop_h = (crop_size, crop_size) if isinstance(crop_size, int) else crop_size
    img_h, img_w = frames[0].shape[:2]
    if kind in ('FiveCrop', 'TenCrop'):
        w_step, h_step = (img_w - crop_w) // 4, (img_h - crop_h) // 4
        offsets = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
    elif kind == 'ThreeCrop':
        if crop_h == img_h:
            w_step = (img_w - crop_w) // 2
            offsets = [(0, 0), (2 * w_step, 0), (w_step, 0)]
        else:
            assert crop_w == img_w
            h_step = (img_h - crop_h) // 2
            offsets = [(0, 0), (0, 2 * h_step), (0, h_step)]
    elif kind == 'CenterCrop':
        offsets = [((img_w - crop_w) // 2, (img_h - crop_h) // 2)]
    else:
        raise KeyError(kind)
    out = []
    for x, y in offsets:
        crop = [f[y:y + crop_h, x:x + crop_w] for f in frames]
        out.extend(crop)
        if kind == 'TenCrop':
            out.extend([np.flip(c, axis=1).copy() for c in crop])
    return out
