"""JPEG decode on the GPU (``bdvcil_amd.decode.JpegDecoder`` -> ``bdv_jpeg_entropy_decode`` on host threads + ``bdv_jpeg_reconstruct_u8``):
bit-equal to the oracle (oracle/jpeg_oracle.py) and to the images Pillow's libjpeg-turbo decodes -- committed fixtures
(tests/golden/jpeg_golden.npz) and fresh streams --, for 4:2:0 / 4:2:2 / 4:4:4 / grey, odd sizes, one- and two-column chroma planes,
restart intervals, mixed quantisation tables and mixed sampling inside one batch, a whole 32 x 8 clip batch at a UCF101 frame's size."""
import io

import numpy as np
import pytest
import torch

from oracle import jpeg_oracle as J
from test_jpeg_cpu import _golden, _picture, _streams

pytestmark = pytest.mark.gpu


def test_golden_images(dev):
    from bdvcil_amd.decode import JpegDecoder
    dec = JpegDecoder(dev, threads=4)
    for data, rgb in _golden():
        out = dec.decode([data])
        assert out.dtype == torch.uint8 and tuple(out.shape) == (1,) + rgb.shape
        assert np.array_equal(out[0].cpu().numpy(), rgb)


def test_fresh_streams_equal_oracle_and_pillow(dev):
    from PIL import Image
    from bdvcil_amd.decode import JpegDecoder
    dec = JpegDecoder(dev, threads=4)
    by_size = {}
    for key, data in _streams():
        by_size.setdefault(key[:2], []).append(data)
    n = 0
    for (h, w), streams in by_size.items():      # twelve streams per size: three samplings x four qualities in ONE call
        out = dec.decode(streams).cpu().numpy()
        assert out.shape == (len(streams), h, w, 3)
        for o, data in zip(out, streams):
            assert np.array_equal(o, J.decode(data)), (h, w)
            assert np.array_equal(o, np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))), (h, w)
            n += 1
    assert n == 96


def test_clip_batch_at_full_size(dev):
    """32 clips x 8 frames of 240 x 320 (BASELINE config 3's batch before Resize): every frame equal to Pillow's decode."""
    from PIL import Image
    from bdvcil_amd.decode import JpegDecoder
    rng = np.random.default_rng(11)
    base = [_picture(240, 320, k % 3, rng) for k in range(6)]
    clips, want = [], []
    for b in range(32):
        clip = []
        for t in range(8):
            frame = np.roll(base[(b + t) % 6], (3 * t, 5 * b), axis=(0, 1))
            buf = io.BytesIO()
            Image.fromarray(frame).save(buf, 'JPEG', quality=60 + (b % 4) * 10, subsampling=2)
            clip.append(buf.getvalue())
            want.append(np.asarray(Image.open(io.BytesIO(clip[-1])).convert('RGB')))
        clips.append(clip)
    out = JpegDecoder(dev, threads=8).decode_clips(clips)
    assert tuple(out.shape) == (32, 8, 240, 320, 3)
    assert np.array_equal(out.cpu().numpy().reshape(256, 240, 320, 3), np.stack(want))


def test_batch_rules(dev):
    from PIL import Image
    from bdvcil_amd.decode import JpegDecoder
    dec = JpegDecoder(dev)
    rng = np.random.default_rng(5)
    a, b = io.BytesIO(), io.BytesIO()
    Image.fromarray(_picture(32, 32, 0, rng)).save(a, 'JPEG')
    Image.fromarray(_picture(32, 48, 0, rng)).save(b, 'JPEG')
    with pytest.raises(ValueError, match='different sizes'):
        dec.decode([a.getvalue(), b.getvalue()])
    with pytest.raises(ValueError, match='empty'):
        dec.decode([])
    with pytest.raises(ValueError, match='lengths'):
        dec.decode_clips([[a.getvalue()], [a.getvalue(), a.getvalue()]])


def test_per_component_scans_on_the_gpu(dev):
    """Non-interleaved scans and 16-bit quantisation tables (streams built by tests/test_jpeg_cpu.py's re-coder): the padding blocks
    of the MCU grid are never coded there, stay zero, and must not reach the picture."""
    from PIL import Image
    from bdvcil_amd.decode import JpegDecoder
    from test_jpeg_cpu import _non_interleaved
    rng = np.random.default_rng(23)
    dec = JpegDecoder(dev, threads=2)
    for (h, w, sub, wide) in ((33, 65, 2, False), (17, 23, 0, True), (24, 40, 1, True)):
        buf = io.BytesIO()
        Image.fromarray(_picture(h, w, 1, rng)).save(buf, 'JPEG', quality=85, subsampling=sub)
        data, _, _ = _non_interleaved(buf.getvalue(), wide)
        want = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        assert np.array_equal(dec.decode([data, buf.getvalue()]).cpu().numpy(), np.stack([want, want]))
