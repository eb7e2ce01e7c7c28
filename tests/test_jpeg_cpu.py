"""JPEG decode, the parts that need no GPU: the oracle (oracle/jpeg_oracle.py) against Pillow's libjpeg-turbo -- its pin -- and
against the committed fixtures; the HOST stage of the product path (``bdv_jpeg_parse`` / ``bdv_jpeg_entropy_decode``: plain C, no
HIP call) against the oracle's header fields and coefficient arrays; the error behaviour."""
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg_golden.npz')


def _golden():
    z = np.load(GOLDEN)
    return [(z[f'stream_{i}'].tobytes(), z[f'rgb_{i}']) for i in range(int(z['n']))]


def _picture(h, w, kind, rng):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 13.0), 128 + 90 * np.cos(xx / 5.0), 128 + 80 * np.sin(yy / 3.0 + xx / 11.0)], -1)
        a = a + rng.normal(0, 12, (h, w, 3))
    elif kind == 1:
        a = rng.integers(0, 256, (h, w, 3)).astype(float)
    else:
        a = np.zeros((h, w, 3))
        a[h // 3:, w // 4:] = (255, 0, 0)
        a[:h // 2, w // 2:] = (0, 255, 255)
        a[::7] = 255
    return np.clip(a, 0, 255).astype(np.uint8)


def _streams():
    """Fresh streams from Pillow: sizes x sampling x quality x content x restart / optimised tables."""
    from PIL import Image
    rng = np.random.default_rng(7)
    for (h, w) in ((16, 16), (17, 23), (33, 65), (8, 8), (1, 1), (5, 3), (100, 7), (120, 160)):
        for sub in (0, 1, 2):
            for q, kind, extra in ((30, 0, {}), (75, 1, {'restart_marker_blocks': 3}), (95, 2, {'restart_marker_rows': 1}), (100, 1, {})):
                buf = io.BytesIO()
                Image.fromarray(_picture(h, w, kind, rng)).save(buf, 'JPEG', quality=q, subsampling=sub, **extra)
                yield (h, w, sub, q, kind), buf.getvalue()


def test_oracle_equals_the_golden_images():
    for data, rgb in _golden():
        assert np.array_equal(J.decode(data), rgb)


def test_oracle_equals_pillow_on_fresh_streams():
    from PIL import Image, features
    assert features.check_feature('libjpeg_turbo')
    n = 0
    for key, data in _streams():
        ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        assert np.array_equal(J.decode(data), ref), key
        n += 1
    assert n == 96


def test_host_stage_equals_the_oracle():
    """Header fields, block grids, quantisation tables and every coefficient of the product path's host stage."""
    from bdvcil_amd.decode import jpeg_entropy_decode, jpeg_parse
    cases = [d for d, _ in _golden()] + [d for k, d in _streams() if k[0] <= 33]
    for data in cases:
        info = J.parse(data)
        geo = J.geometry(info)
        want = J.entropy_decode(data, info)
        got_info, got = jpeg_entropy_decode(data)
        assert (got_info.width, got_info.height, got_info.ncomp) == (info['width'], info['height'], len(info['comps']))
        off = 0
        for c, (comp, g, w) in enumerate(zip(info['comps'], geo['comps'], want)):
            assert (got_info.blocks_w[c], got_info.blocks_h[c], got_info.down_w[c], got_info.down_h[c]) == (g['bw'], g['bh'], g['dw'], g['dh'])
            assert got_info.coef_offset[c] == off
            assert np.array_equal(np.ctypeslib.as_array(got_info.qt)[c], info['qt'][comp['tq']].astype(np.uint16))
            assert np.array_equal(got[off:off + w.size].reshape(w.shape), w)
            off += w.size
        assert got_info.coef_count == off
        assert jpeg_parse(data).geometry_key() == got_info.geometry_key()


def test_errors_are_reported_not_guessed():
    from PIL import Image
    from bdvcil_amd._lib import HipExtensionError
    from bdvcil_amd.decode import jpeg_entropy_decode, jpeg_parse
    rng = np.random.default_rng(3)
    img = Image.fromarray(_picture(32, 32, 0, rng))
    buf = io.BytesIO()
    img.save(buf, 'JPEG', progressive=True)
    with pytest.raises((HipExtensionError, RuntimeError), match='progressive|SOF'):
        jpeg_parse(buf.getvalue())
    with pytest.raises(ValueError):
        J.decode(buf.getvalue())
    with pytest.raises((HipExtensionError, RuntimeError), match='SOI'):
        jpeg_parse(b'\x89PNG\r\n\x1a\n' + bytes(32))
    buf = io.BytesIO()
    img.convert('CMYK').save(buf, 'JPEG')
    with pytest.raises((HipExtensionError, RuntimeError), match='components'):
        jpeg_parse(buf.getvalue())
    # a stream of another geometry than the buffers were sized for
    a, b = io.BytesIO(), io.BytesIO()
    img.save(a, 'JPEG', subsampling=2)
    img.resize((48, 32)).save(b, 'JPEG', subsampling=2)
    info = jpeg_parse(a.getvalue())
    with pytest.raises((HipExtensionError, RuntimeError), match='geometry'):
        jpeg_entropy_decode(b.getvalue(), info, np.empty(info.coef_count, dtype=np.int16))
    # a truncated entropy segment decodes to the end with zero bits (as libjpeg does, with a warning there): no crash, right size
    data = a.getvalue()
    _, co = jpeg_entropy_decode(data[:len(data) - 40])
    assert co.size == info.coef_count


def test_batch_host_stage_equals_per_image_calls():
    """``bdv_jpeg_entropy_decode_batch`` (std::thread workers inside the library) == one ``bdv_jpeg_entropy_decode`` per stream; the
    first failing stream's error comes back with its index."""
    import ctypes
    from PIL import Image
    from bdvcil_amd._lib import HipExtensionError, check, lib
    from bdvcil_amd.decode import jpeg_entropy_decode, jpeg_parse
    rng = np.random.default_rng(9)
    streams = []
    for i in range(13):
        buf = io.BytesIO()
        Image.fromarray(_picture(40, 56, i % 3, rng)).save(buf, 'JPEG', quality=50 + 4 * i, subsampling=2)
        streams.append(buf.getvalue())
    info = jpeg_parse(streams[0])

    def run(ss, threads):
        n = len(ss)
        coefs = np.full((n, info.coef_count), 7, dtype=np.int16)
        qts = np.zeros((n, 3, 64), dtype=np.uint16)
        ptrs = (ctypes.c_char_p * n)(*ss)
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in ss])
        check(lib().bdv_jpeg_entropy_decode_batch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), n, ctypes.byref(info),
                                                  coefs.ctypes.data, qts.ctypes.data, threads), 'bdv_jpeg_entropy_decode_batch')
        return coefs, qts
    for threads in (1, 4, 32):
        coefs, qts = run(streams, threads)
        for i, s in enumerate(streams):
            inf, c = jpeg_entropy_decode(s)
            assert np.array_equal(coefs[i], c) and np.array_equal(qts[i], np.ctypeslib.as_array(inf.qt))
    buf = io.BytesIO()
    Image.fromarray(_picture(40, 64, 0, rng)).save(buf, 'JPEG')
    with pytest.raises((HipExtensionError, RuntimeError), match='image 5.*geometry'):
        run(streams[:5] + [buf.getvalue()] + streams[5:], 3)
