"""CPU restatement of the baseline-JPEG decode the reference's frame pipeline starts with.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path never does.

Reference call site: the train / val / test pipelines open with ``RawFrameDecode`` (configs/ucf101/bgmix_plus_randAug/
bgmix_seed_1000_inc_10_stages_bgmix_plus_randAug.py:124-141; UPSTREAM mmaction2 ``RawFrameDecode`` -> ``mmcv.imfrombytes(...,
channel_order='rgb')`` -> ``cv2.imdecode``), i.e. the algorithm lives in a third-party dependency that is absent from /root/reference
AND from this image: OpenCV's bundled libjpeg(-turbo) with its defaults -- ISLOW inverse DCT, "fancy" (triangle) chroma upsampling,
fixed-point YCbCr -> RGB.  Those three steps are specified to the bit by the IJG / libjpeg-turbo sources (jidctint.c, jdsample.c,
jdcolor.c; the SIMD forms are bit-identical by design), restated here from the published algorithm.

Pin: Pillow 12.2 links libjpeg-turbo (``PIL.features``: jpg 6.2, libjpeg_turbo True) and decodes with the same defaults;
tests/test_jpeg_cpu.py holds this oracle bit-equal to ``PIL.Image.open(...).convert('RGB')`` on streams Pillow itself wrote in this
container (4:2:0 / 4:2:2 / 4:4:4 / grey, odd sizes, restart intervals, optimised tables, qualities 30 - 100) and on the committed
fixtures of tests/golden/jpeg_golden.npz (made by tests/golden/make_golden_jpeg.py).  So: pinned by the library family the
reference's decoder comes from, not by a run of the reference's own decode call (cv2 / mmcv are not importable).

Scope: baseline sequential DCT (SOF0 / SOF1 Huffman, 8-bit), 1 or 3 components, luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma,
interleaved or per-component scans, DRI / RSTn.  Progressive, arithmetic-coded, CMYK and 12-bit streams raise ``ValueError``."""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63], dtype=np.int64)      # jutils.c jpeg_natural_order: zigzag position -> natural (row-major) index


class _Bits:
    """Entropy-coded segment reader: MSB first, 0xFF00 -> 0xFF, stops at a marker (jdhuff.c jpeg_fill_bit_buffer)."""

    def __init__(self, data: bytes, pos: int):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        while self.n <= 24:
            if self.p < len(self.d):
                b = self.d[self.p]
                if b == 0xFF:
                    nxt = self.d[self.p + 1] if self.p + 1 < len(self.d) else 0xD9
                    if nxt == 0x00:
                        self.p += 2
                    else:                 # a marker: feed zeros, do not advance (libjpeg does the same and warns)
                        b = 0
                else:
                    self.p += 1
            else:
                b = 0
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8

    def get(self, k: int) -> int:
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def restart(self):
        """Byte-align, expect RSTn at the read position, skip it."""
        self.acc, self.n = 0, 0
        while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


class _Huff:
    """JPEG Annex C code construction + Annex F.2.2.3 decoding (jdhuff.c jpeg_make_d_derived_tbl, slow path)."""

    def __init__(self, counts, symbols):
        self.maxcode, self.valptr, self.mincode = [-1] * 18, [0] * 17, [0] * 17
        self.sym = list(symbols)
        code, k = 0, 0
        for length in range(1, 17):
            self.valptr[length] = k
            self.mincode[length] = code
            code += counts[length - 1]
            k += counts[length - 1]
            self.maxcode[length] = code - 1 if counts[length - 1] else -1
            code <<= 1
        self.maxcode[17] = 1 << 30

    def decode(self, bits: _Bits) -> int:
        code, length = bits.get(1), 1
        while code > self.maxcode[length]:
            code = (code << 1) | bits.get(1)
            length += 1
            if length > 16:
                return 0
        return self.sym[self.valptr[length] + code - self.mincode[length]]


def _extend(v: int, s: int) -> int:
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1      # F.2.2.1 EXTEND


def parse(data: bytes) -> dict:
    """Markers up to EOI -> {'width','height','comps':[{'id','h','v','tq'}], 'qt':{tq: (64,) natural order}, 'scans':[...]}."""
    if data[:2] != b'\xff\xd8':
        raise ValueError('not a JPEG stream (no SOI)')
    info = {'qt': {}, 'dc': {}, 'ac': {}, 'scans': [], 'ri': 0}
    p = 2
    while p < len(data):
        if data[p] != 0xFF:
            p += 1
            continue
        m = data[p + 1]
        p += 2
        if m == 0xD9:
            break
        if m in (0x01, 0xFF) or 0xD0 <= m <= 0xD7:
            continue
        L = (data[p] << 8) | data[p + 1]
        seg = data[p + 2:p + L]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                if pq:
                    vals = [(seg[q + 1 + 2 * i] << 8) | seg[q + 2 + 2 * i] for i in range(64)]
                    q += 129
                else:
                    vals = list(seg[q + 1:q + 65])
                    q += 65
                t = np.zeros(64, dtype=np.int64)
                t[ZIGZAG] = vals
                info['qt'][tq] = t
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise ValueError('only 8-bit samples')
            info['height'], info['width'] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            nc = seg[5]
            if nc not in (1, 3):
                raise ValueError(f'{nc} components: only grey and YCbCr')
            info['comps'] = [{'id': seg[6 + 3 * i], 'h': seg[7 + 3 * i] >> 4, 'v': seg[7 + 3 * i] & 15, 'tq': seg[8 + 3 * i]} for i in range(nc)]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise ValueError(f'SOF marker 0x{m:02x}: only baseline / extended sequential Huffman streams')
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                (info['ac'] if tc else info['dc'])[th] = _Huff(counts, seg[q + 17:q + 17 + n])
                q += 17 + n
        elif m == 0xDD:
            info['ri'] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            sel = [(seg[1 + 2 * i], seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)]
            if (seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]) != (0, 63, 0):
                raise ValueError('spectral selection / successive approximation: not a sequential scan')
            # the tables in force NOW belong to this scan (they may be redefined before the next one)
            info['scans'].append({'sel': sel, 'pos': p + L, 'dc': dict(info['dc']), 'ac': dict(info['ac']), 'ri': info['ri']})
            p += L
            while p + 1 < len(data) and not (data[p] == 0xFF and data[p + 1] != 0 and not 0xD0 <= data[p + 1] <= 0xD7):
                p += 1
            continue
        p += L
    if 'comps' not in info or not info['scans']:
        raise ValueError('no frame header / no scan')
    return info


def geometry(info: dict) -> dict:
    """Block grids as jdmaster.c / jdinput.c lay them out: per component the padded grid (whole MCUs) and the real sample size."""
    comps = info['comps']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    if len(comps) == 3 and not (comps[1]['h'] == comps[2]['h'] == 1 and comps[1]['v'] == comps[2]['v'] == 1 and
                                (comps[0]['h'], comps[0]['v']) in ((1, 1), (2, 1), (2, 2))):
        raise ValueError('sampling factors: only 4:4:4, 4:2:2 and 4:2:0')
    W, H = info['width'], info['height']
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    out = {'hmax': hmax, 'vmax': vmax, 'mcux': mcux, 'mcuy': mcuy, 'comps': []}
    for c in comps:
        out['comps'].append({'bw': mcux * c['h'], 'bh': mcuy * c['v'],                                 # blocks, padded to whole MCUs
                             'dw': -(-W * c['h'] // hmax), 'dh': -(-H * c['v'] // vmax)})              # downsampled_width / _height
    return out


def entropy_decode(data: bytes, info: dict) -> list:
    """-> per component an int16 array (bh, bw, 64) of QUANTISED coefficients in natural order (jdhuff.c decode_mcu)."""
    geo = geometry(info)
    comps = info['comps']
    coefs = [np.zeros((g['bh'], g['bw'], 64), dtype=np.int16) for g in geo['comps']]
    index = {c['id']: i for i, c in enumerate(comps)}
    for scan in info['scans']:
        bits = _Bits(data, scan['pos'])
        ids = [index[s[0]] for s in scan['sel']]
        pred = [0] * len(comps)
        if len(ids) > 1:
            units = [(mx, my) for my in range(geo['mcuy']) for mx in range(geo['mcux'])]
        else:       # a single-component scan runs over the component's own (unpadded) block grid, A.2.3
            g = geo['comps'][ids[0]]
            units = [(bx, by) for by in range(-(-g['dh'] // 8)) for bx in range(-(-g['dw'] // 8))]
        for u, (ux, uy) in enumerate(units):
            if scan['ri'] and u and u % scan['ri'] == 0:
                bits.restart()
                pred = [0] * len(comps)
            for (cid, td, ta), ci in zip(scan['sel'], ids):
                c = comps[ci]
                blocks = [(ux * c['h'] + bx, uy * c['v'] + by) for by in range(c['v']) for bx in range(c['h'])] if len(ids) > 1 else [(ux, uy)]
                for (bx, by) in blocks:
                    blk = coefs[ci][by, bx]
                    s = scan['dc'][td].decode(bits)
                    if s:
                        pred[ci] += _extend(bits.get(s), s)
                    blk[0] = np.int16(pred[ci])
                    k = 1
                    while k < 64:
                        rs = scan['ac'][ta].decode(bits)
                        r, s = rs >> 4, rs & 15
                        if s:
                            k += r
                            if k > 63:
                                break
                            blk[ZIGZAG[k]] = np.int16(_extend(bits.get(s), s))
                            k += 1
                        elif r == 15:
                            k += 16
                        else:
                            break
    return coefs


# ---- jidctint.c (ISLOW), CONST_BITS = 13, PASS1_BITS = 2 ----------------------------------------------------------------------
_F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069,
          f2_053=16819, f2_562=20995, f3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """One pass over axis -1 of (..., 8) int64; jidctint.c jpeg_idct_islow, either pass (they differ in the final shift only)."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * _F['f0_541']
    tmp2 = z1 + z3 * (-_F['f1_847'])
    tmp3 = z1 + z2 * _F['f0_765']
    z2, z3 = v[..., 0], v[..., 4]
    tmp0, tmp1 = (z2 + z3) << 13, (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * _F['f1_175']
    tmp0, tmp1, tmp2, tmp3 = tmp0 * _F['f0_298'], tmp1 * _F['f2_053'], tmp2 * _F['f3_072'], tmp3 * _F['f1_501']
    z1, z2, z3, z4 = z1 * (-_F['f0_899']), z2 * (-_F['f2_562']), z3 * (-_F['f1_961']) + z5, z4 * (-_F['f0_390']) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return np.stack([_descale(tmp10 + tmp3, shift), _descale(tmp11 + tmp2, shift), _descale(tmp12 + tmp1, shift), _descale(tmp13 + tmp0, shift),
                     _descale(tmp13 - tmp0, shift), _descale(tmp12 - tmp1, shift), _descale(tmp11 - tmp2, shift), _descale(tmp10 - tmp3, shift)], axis=-1)


def _range_limit_idct(x):
    """sample_range_limit + CENTERJSAMPLE indexed with x & RANGE_MASK (jdmaster.c prepare_range_limit_table)."""
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_plane(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """(bh, bw, 64) quantised coefficients -> (8 bh, 8 bw) uint8 samples."""
    bh, bw, _ = coef.shape
    blk = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(bh, bw, 8, 8)           # [row][col]
    ws = _idct_1d(blk.swapaxes(-1, -2), 13 - 2).swapaxes(-1, -2)                        # pass 1: down the columns
    out = _idct_1d(ws, 13 + 2 + 3)                                                      # pass 2: along the rows
    return _range_limit_idct(out).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


# ---- jdsample.c fancy upsampling ---------------------------------------------------------------------------------------------
def _h2v1_fancy(row: np.ndarray) -> np.ndarray:
    """(rows, dw) -> (rows, 2 dw): 3/4 nearer + 1/4 further, rounding 1 / 2 alternately; the end columns are copies."""
    s = row.astype(np.int64)
    left, right = np.concatenate([s[:, :1], s[:, :-1]], axis=1), np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
    even, odd = (3 * s + left + 1) >> 2, (3 * s + right + 2) >> 2
    even[:, 0], odd[:, -1] = s[:, 0], s[:, -1]
    out = np.empty((s.shape[0], 2 * s.shape[1]), dtype=np.int64)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out


def _h2v2_fancy(plane: np.ndarray) -> np.ndarray:
    """(dh, dw) -> (2 dh, 2 dw): rows 3/4 this + 1/4 nearer neighbour (the edge rows see themselves), then the same along x on the
    column sums, with rounding 8 / 7 alternately (h2v2_fancy_upsample)."""
    s = plane.astype(np.int64)
    up, down = np.concatenate([s[:1], s[:-1]]), np.concatenate([s[1:], s[-1:]])
    out = np.empty((2 * s.shape[0], 2 * s.shape[1]), dtype=np.int64)
    for v, other in ((0, up), (1, down)):
        cs = 3 * s + other
        last, nxt = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1), np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        even, odd = (3 * cs + last + 8) >> 4, (3 * cs + nxt + 7) >> 4
        even[:, 0], odd[:, -1] = (4 * cs[:, 0] + 8) >> 4, (4 * cs[:, -1] + 7) >> 4
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out


def upsample(plane: np.ndarray, h: int, v: int, dw: int, dh: int) -> np.ndarray:
    """Chroma plane (padded) -> (v dh, h dw) at luma resolution; only the real dw x dh samples take part (jdmainct.c context rows)."""
    p = plane[:dh, :dw]
    if h == 1 and v == 1:
        return p.astype(np.int64)
    if dw <= 2:      # jinit_upsampler: the fancy forms need more than two columns, else plain replication
        return np.repeat(np.repeat(p, v, axis=0), h, axis=1).astype(np.int64)
    if h == 2 and v == 1:
        return _h2v1_fancy(p)
    if h == 2 and v == 2:
        return _h2v2_fancy(p)
    raise ValueError(f'upsampling {h}x{v}')


# ---- jdcolor.c ycc_rgb_convert -----------------------------------------------------------------------------------------------
def _fix(x):
    return int(x * (1 << 16) + 0.5)


_X = np.arange(256, dtype=np.int64) - 128
CR_R = (_fix(1.40200) * _X + (1 << 15)) >> 16
CB_B = (_fix(1.77200) * _X + (1 << 15)) >> 16
CR_G = -_fix(0.71414) * _X
CB_G = -_fix(0.34414) * _X + (1 << 15)


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64), cr.astype(np.int64)
    r = y + CR_R[cr]
    g = y + ((CB_G[cb] + CR_G[cr]) >> 16)
    b = y + CB_B[cb]
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode_from_coefs(info: dict, coefs: list) -> np.ndarray:
    geo = geometry(info)
    W, H = info['width'], info['height']
    planes = [idct_plane(c, info['qt'][comp['tq']]) for c, comp in zip(coefs, info['comps'])]
    if len(planes) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], axis=-1)
    hmax, vmax = geo['hmax'], geo['vmax']
    up = [upsample(p, hmax // c['h'], vmax // c['v'], g['dw'], g['dh'])[:H, :W] for p, c, g in zip(planes, info['comps'], geo['comps'])]
    return ycc_to_rgb(up[0], up[1], up[2])


def decode(data: bytes) -> np.ndarray:
    """JPEG bytes -> (H, W, 3) uint8 RGB, as libjpeg-turbo's defaults give it."""
    info = parse(data)
    return decode_from_coefs(info, entropy_decode(data, info))
