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


# ---- streams Pillow cannot write: per-component (non-interleaved) scans, 16-bit quantisation tables -----------------------------
def _huff_code_table(counts, symbols):
    """Annex C: symbol -> (code, length)."""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[symbols[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


class _BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)      # pad with ones
        return bytes(self.out)


def _encode_blocks(blocks, dc_tab, ac_tab):
    """Baseline Huffman coding (F.1.2) of a sequence of natural-order coefficient blocks of ONE component."""
    w, pred = _BitWriter(), 0
    for blk in blocks:
        zz = [int(blk[J.ZIGZAG[k]]) for k in range(64)]
        diff = zz[0] - pred
        pred = zz[0]
        s = 0 if diff == 0 else int(abs(diff)).bit_length()
        w.put(*dc_tab[s])
        if s:
            w.put(diff if diff > 0 else diff + (1 << s) - 1, s)
        run = 0
        last = max([k for k in range(1, 64) if zz[k] != 0], default=0)
        for k in range(1, last + 1):
            if zz[k] == 0:
                run += 1
                continue
            while run > 15:
                w.put(*ac_tab[0xF0])
                run -= 16
            s = int(abs(zz[k])).bit_length()
            w.put(*ac_tab[(run << 4) | s])
            w.put(zz[k] if zz[k] > 0 else zz[k] + (1 << s) - 1, s)
            run = 0
        if last < 63:
            w.put(*ac_tab[0x00])
    return w.flush()


def _segments(data):
    """(marker, payload) of every segment before the first SOS, plus the parsed tables."""
    p, out = 2, []
    while True:
        m = data[p + 1]
        L = (data[p + 2] << 8) | data[p + 3]
        if m == 0xDA:
            return out
        out.append((m, data[p + 4:p + 2 + L]))
        p += 2 + L


def _non_interleaved(data, wide_tables=False):
    """Re-codes a one-scan 4:2:0 / 4:4:4 stream from Pillow as three one-component scans over each component's own block grid
    (A.2.3), optionally with the quantisation tables rewritten in their 16-bit form (Pq = 1)."""
    info = J.parse(data)
    coefs = J.entropy_decode(data, info)
    geo = J.geometry(info)
    out = bytearray(b'\xff\xd8')
    tabs = {}
    for m, payload in _segments(data):
        if m == 0xC4:
            q = 0
            while q < len(payload):
                tc, th = payload[q] >> 4, payload[q] & 15
                counts = list(payload[q + 1:q + 17])
                n = sum(counts)
                tabs[(tc, th)] = _huff_code_table(counts, list(payload[q + 17:q + 17 + n]))
                q += 17 + n
        if m == 0xDB and wide_tables:
            q, new = 0, bytearray()
            while q < len(payload):
                tq = payload[q] & 15
                new.append(0x10 | tq)
                for v in payload[q + 1:q + 65]:
                    new += bytes([0, v])
                q += 65
            payload = bytes(new)
        if 0xE0 <= m <= 0xEF:
            continue
        out += bytes([0xFF, m]) + (len(payload) + 2).to_bytes(2, 'big') + payload
    sel = {c['id']: (0, 0) if i == 0 else (1, 1) for i, c in enumerate(info['comps'])}       # Pillow: tables 0 for luma, 1 for chroma
    for ci, (c, g) in enumerate(zip(info['comps'], geo['comps'])):
        nbx, nby = -(-g['dw'] // 8), -(-g['dh'] // 8)
        blocks = [coefs[ci][by, bx] for by in range(nby) for bx in range(nbx)]
        td, ta = sel[c['id']]
        out += b'\xff\xda' + (8).to_bytes(2, 'big') + bytes([1, c['id'], (td << 4) | ta, 0, 63, 0])
        out += _encode_blocks(blocks, tabs[(0, td)], tabs[(1, ta)])
    return bytes(out + b'\xff\xd9'), info, coefs


def test_per_component_scans_and_wide_tables():
    """Streams no Pillow option produces, built by re-coding Pillow's coefficients: three one-component scans (each over its own
    block grid, without the MCU padding) and 16-bit quantisation tables.  Pillow / libjpeg-turbo decodes them too: the oracle, the
    host stage and Pillow must agree with each other and with the interleaved original."""
    from PIL import Image
    from bdvcil_amd.decode import jpeg_entropy_decode
    rng = np.random.default_rng(17)
    for (h, w, sub, wide) in ((33, 65, 2, False), (17, 23, 0, False), (40, 56, 2, True), (24, 40, 1, True)):
        buf = io.BytesIO()
        Image.fromarray(_picture(h, w, 0, rng)).save(buf, 'JPEG', quality=80, subsampling=sub)
        orig = buf.getvalue()
        data, info, coefs = _non_interleaved(orig, wide)
        ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        assert np.array_equal(ref, np.asarray(Image.open(io.BytesIO(orig)).convert('RGB')))      # the re-coding kept every coefficient
        assert np.array_equal(J.decode(data), ref)
        assert len(J.parse(data)['scans']) == 3
        got_info, got = jpeg_entropy_decode(data)
        geo = J.geometry(info)
        off = 0
        for ci, g in enumerate(geo['comps']):
            nbx, nby = -(-g['dw'] // 8), -(-g['dh'] // 8)
            want = np.zeros_like(coefs[ci])
            want[:nby, :nbx] = coefs[ci][:nby, :nbx]              # blocks of the MCU padding are not coded in a one-component scan
            assert np.array_equal(got[off:off + want.size].reshape(want.shape), want), ci
            off += want.size
