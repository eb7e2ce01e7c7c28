"""Writes tests/golden/jpeg_golden.npz: JPEG streams produced by Pillow (libjpeg-turbo) in this container and the RGB images the same
library decodes them to -- the pin of oracle/jpeg_oracle.py and of the HIP decode (tests/test_jpeg_cpu.py, tests/test_jpeg_gpu.py).
The reference holds no JPEG fixture and its decoder (cv2 through mmcv) is not importable here; Pillow's libjpeg-turbo is the same
decoder family with the same defaults.  Run from the repo root:  python tests/golden/make_golden_jpeg.py"""
import io
import os

import numpy as np
from PIL import Image, features

assert features.check_feature('libjpeg_turbo'), 'the fixtures are meant to come from libjpeg-turbo'
rng = np.random.default_rng(2026)


def picture(h, w, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 'smooth':
        a = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 13.0), 128 + 90 * np.cos(xx / 5.0), 128 + 80 * np.sin(yy / 3.0 + xx / 11.0)], -1)
        a = a + rng.normal(0, 12, (h, w, 3))
    elif kind == 'noise':
        a = rng.integers(0, 256, (h, w, 3)).astype(float)
    else:      # saturated edges: exercises the range limits of the inverse DCT and of the colour conversion
        a = np.zeros((h, w, 3))
        a[h // 3:, w // 4:] = (255, 0, 0)
        a[:h // 2, w // 2:] = (0, 255, 255)
        a[::7] = 255
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


CASES = [  # (h, w, kind, save options)
    (240, 320, 'smooth', dict(quality=75, subsampling=2)),                          # a UCF101 frame's geometry, 4:2:0
    (72, 104, 'noise', dict(quality=95, subsampling=2, restart_marker_rows=1)),
    (33, 65, 'edges', dict(quality=90, subsampling=2)),
    (33, 65, 'smooth', dict(quality=60, subsampling=1)),                            # 4:2:2
    (17, 23, 'noise', dict(quality=100, subsampling=0)),                            # 4:4:4
    (17, 23, 'edges', dict(quality=30, subsampling=2, restart_marker_blocks=3)),
    (64, 48, 'smooth', dict(quality=85, subsampling=2, optimize=True)),
    (5, 3, 'noise', dict(quality=80, subsampling=2)),                               # two chroma columns: plain replication
    (1, 1, 'noise', dict(quality=80, subsampling=2)),
    (40, 56, 'grey', dict(quality=80)),
]
out = {}
for i, (h, w, kind, opts) in enumerate(CASES):
    img = picture(h, w, 'smooth').convert('L') if kind == 'grey' else picture(h, w, kind)
    buf = io.BytesIO()
    img.save(buf, 'JPEG', **opts)
    data = buf.getvalue()
    out[f'stream_{i}'] = np.frombuffer(data, dtype=np.uint8)
    out[f'rgb_{i}'] = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
out['n'] = np.int64(len(CASES))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jpeg_golden.npz')
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), 'bytes')
