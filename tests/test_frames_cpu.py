"""Frame-pipeline pieces that need no GPU: the resize oracle's sanity properties (it is UNPINNED -- cv2 is absent, oracle/resize_oracle.py
says so), and the host-side draws of the product path (``sample_frames``, ``rescale_size``, ``MultiScaleCropResize.draw``) against the
oracle's restatement under the same seeds."""
import random

import numpy as np

from oracle import resize_oracle as R


def _float_bilinear(img, Wd, Hd):
    Hs, Ws = img.shape[:2]
    a = img.astype(np.float64)
    fx, fy = (np.arange(Wd) + 0.5) * Ws / Wd - 0.5, (np.arange(Hd) + 0.5) * Hs / Hd - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    wx, wy = fx - x0, fy - y0
    x0c, x1c, y0c, y1c = np.clip(x0, 0, Ws - 1), np.clip(x0 + 1, 0, Ws - 1), np.clip(y0, 0, Hs - 1), np.clip(y0 + 1, 0, Hs - 1)
    top = a[y0c][:, x0c] * (1 - wx)[None, :, None] + a[y0c][:, x1c] * wx[None, :, None]
    bot = a[y1c][:, x0c] * (1 - wx)[None, :, None] + a[y1c][:, x1c] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


def test_resize_oracle_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 80, 3)).astype(np.uint8)
    assert np.array_equal(R.resize_linear_u8(img, 80, 60), img)                                  # same size: a copy
    for (Wd, Hd) in ((85, 64), (56, 56), (33, 21), (160, 120), (79, 59), (200, 37)):
        out = R.resize_linear_u8(img, Wd, Hd)
        assert out.shape == (Hd, Wd, 3) and out.dtype == np.uint8
        assert np.abs(out - _float_bilinear(img, Wd, Hd)).max() < 1.0                            # fixed point: within one grey level
    const = np.full((50, 70, 3), 137, np.uint8)
    for (Wd, Hd) in ((224, 224), (35, 25), (71, 49)):
        assert (R.resize_linear_u8(const, Wd, Hd) == 137).all()
    half = R.resize_linear_u8(img, 40, 30)                                                      # exact 2x shrink: the 2x2 box mean
    a = img.astype(int)
    assert np.array_equal(half, ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))


def test_rescale_size():
    from bdvcil_amd.decode import rescale_size
    for (w, h) in ((320, 240), (240, 320), (341, 256), (100, 100), (1280, 720), (17, 333)):
        assert rescale_size(w, h, (-1, 256)) == R.rescale_size(w, h, (-1, 256))
        assert min(rescale_size(w, h, (-1, 256))) == 256
    assert rescale_size(320, 240, (-1, 256)) == (341, 256)


def test_sample_frames_follow_the_oracle():
    from bdvcil_amd.decode import sample_frames
    for total in (1, 3, 7, 8, 9, 15, 16, 100, 301):
        for test_mode in (False, True):
            np.random.seed(total)
            got = sample_frames(total, 8, test_mode=test_mode)
            np.random.seed(total)
            want = R.sample_frames(total, 8, test_mode=test_mode, rng=np.random)
            assert np.array_equal(got, want), (total, test_mode)
            assert got.shape == (8,) and got.min() >= 1 and got.max() <= total
    assert np.array_equal(sample_frames(100, 8, test_mode=True), [7, 19, 32, 44, 57, 69, 82, 94])   # segment centres


def test_multi_scale_crop_draws_follow_the_oracle():
    from bdvcil_amd.frontend import MultiScaleCropResize
    for nfc in (5, 13):
        for rc in (False, True):
            m = MultiScaleCropResize(224, (1, 0.875, 0.75, 0.66), 1, rc, nfc)
            random.seed(10 * nfc + rc)
            got = [m.draw(341, 256) for _ in range(50)]
            random.seed(10 * nfc + rc)
            want = [R.multi_scale_crop_box(341, 256, (224, 224), (1, 0.875, 0.75, 0.66), 1, rc, nfc, rng=random) for _ in range(50)]
            assert got == want
            for (x, y, w, h) in got:
                assert 0 <= x and x + w <= 341 and 0 <= y and y + h <= 256 and w in (256, 224, 192, 168) and h in (256, 224, 192, 168)
