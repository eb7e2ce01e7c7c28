"""RandAugment stage, CPU side: the numpy oracle against the golden vectors produced by the reference's own
rand_augment.py, against Pillow itself on fresh frames, and the product's host-side draw / parameter logic (no kernels)."""
import json
import os
import random

import numpy as np
import pytest

from oracle import augment_oracle as AO

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'randaug_golden.npz')


@pytest.fixture(scope='module')
def golden():
    z = np.load(GOLDEN)
    return z, json.loads(str(z['cases'])), json.loads(str(z['calls']))


def test_oracle_matches_reference_golden_per_operation(golden):
    z, cases, _ = golden
    seen = set()
    for i, c in enumerate(cases):
        got = AO.apply_op(c['name'], z['imgs'][c['img']], c['val'], c['flip'], tuple(c['loc']))
        assert np.array_equal(got, z[f'op{i}']), (i, c)
        seen.add(c['name'])
    assert seen == {n for n, _, _ in AO.OP_TABLE} and len(cases) >= 140


def test_oracle_matches_reference_golden_whole_calls(golden):
    z, _, calls = golden
    drawn = set()
    for c in calls:
        random.seed(c['seed'])
        np.random.seed(c['seed'])
        frames, flag, names = AO.rand_augment([z['imgs'][k].copy() for k in c['imgs']], 2, 10, 0.75)
        assert flag == c['randAug']
        assert np.array_equal(np.stack(frames), z[f'call{c["seed"]}']), c
        drawn.update(names)
    assert len(drawn) >= 12                       # the 40 seeds reach most of the table


def test_oracle_matches_pillow_on_fresh_frames():
    PIL = pytest.importorskip('PIL')
    from PIL import Image, ImageDraw, ImageEnhance, ImageOps
    rng = np.random.default_rng(7)
    for (H, W) in [(256, 340), (37, 53)]:
        for img in (rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
                    rng.integers(30, 180, (H // 8 + 1, W // 8 + 1, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W].copy()):
            P = Image.fromarray(img)
            fill = AO.FILL_COLOR
            checks = [
                (AO.autocontrast(img), ImageOps.autocontrast(P)),
                (AO.equalize(img), ImageOps.equalize(P)),
                (AO.solarize(img, 85.33333333333333), ImageOps.solarize(P, 85.33333333333333)),
                (AO.posterize(img, 5.33), ImageOps.posterize(P, 5)),
                (AO.color(img, 0.35), ImageEnhance.Color(P).enhance(0.35)),
                (AO.contrast(img, 0.35), ImageEnhance.Contrast(P).enhance(0.35)),
                (AO.brightness(img, 0.35), ImageEnhance.Brightness(P).enhance(0.35)),
                (AO.sharpness(img, 0.35), ImageEnhance.Sharpness(P).enhance(0.35)),
                (AO.affine_nearest(img, (1, 0.1, 0, 0, 1, 0)), P.transform(P.size, Image.AFFINE, (1, 0.1, 0, 0, 1, 0), fillcolor=fill)),
                (AO.affine_nearest(img, (1, 0, 0, -0.1, 1, 0)), P.transform(P.size, Image.AFFINE, (1, 0, 0, -0.1, 1, 0), fillcolor=fill)),
                (AO.affine_nearest(img, (1, 0, 0.1 * W, 0, 1, 0)), P.transform(P.size, Image.AFFINE, (1, 0, 0.1 * W, 0, 1, 0), fillcolor=fill)),
                (AO.affine_nearest(img, (1, 0, 0, 0, 1, -0.1 * H)), P.transform(P.size, Image.AFFINE, (1, 0, 0, 0, 1, -0.1 * H), fillcolor=fill)),
                (AO.rotate(img, 10.0), P.rotate(10.0, fillcolor=fill)),
                (AO.rotate(img, -10.0), P.rotate(-10.0, fillcolor=fill)),
            ]
            for k, (mine, ref) in enumerate(checks):
                assert np.array_equal(mine, np.array(ref)), (H, W, k)
            loc = (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
            x0, y0 = int(max(0, loc[0] - 37.33 / 2.)), int(max(0, loc[1] - 37.33 / 2.))
            Q = P.copy()
            ImageDraw.Draw(Q).rectangle((x0, y0, min(W, x0 + 37.33), min(H, y0 + 37.33)), fill)
            assert np.array_equal(AO.cutout_abs(img, 37.33, loc), np.array(Q))


def test_host_draws_and_rows_mirror_the_reference(golden):
    """bdvcil_amd.augment.RandAugment draws what the reference draws (flags per seed from the golden file) and encodes
    the operations into the device tables the C ABI documents.  No kernel is launched."""
    from bdvcil_amd import augment as A
    z, _, calls = golden
    aug = A.RandAugment(2, 10, 0.75)
    assert [n for n, _, _ in aug.augment_list] == [n for n, _, _ in AO.OP_TABLE]
    for c in calls:
        random.seed(c['seed'])
        np.random.seed(c['seed'])
        d = aug.draw(24, 36)
        random.seed(c['seed'])
        np.random.seed(c['seed'])
        _, flag, names = AO.rand_augment([z['imgs'][0]], 2, 10, 0.75)
        assert (d is not None) == c['randAug'] == flag
        if d is not None:
            assert [o[0] for o in d[0]] == names and 1.0 <= d[2][0] <= 36 and 1.0 <= d[2][1] <= 24   # uniform(W): low=W, high=1
    # table encoding
    H, W = 256, 340
    ri, rd = A.op_row('ShearX', 0.1, True, (0, 0), H, W)
    assert ri[0] == A.AFFINE_FIXED and ri[1] == 65536 and ri[2] == AO._fix(-0.1) and ri[3] == AO._fix(0.5 - 0.05) and ri[7] == 0x7C7468
    ri, rd = A.op_row('TranslateX', 0.1, False, (0, 0), H, W)
    assert ri[0] == A.AFFINE_SCALE and rd == [1.0, 0.1 * W, 1.0, 0.0]
    ri, rd = A.op_row('Posterize', 4 + 4 / 3, False, (0, 0), H, W)
    assert ri[:2] == [A.POSTERIZE, 5]
    ri, rd = A.op_row('CutoutAbs', 112 / 3, False, (300.7, 10.2), H, W)
    assert ri[:5] == [A.CUTOUT, 282, 0, 319, 37]
    ri, rd = A.op_row('Rotate', 0.0, False, (0, 0), H, W)
    assert ri[0] == A.IDENTITY
    with pytest.raises(KeyError):
        A.op_row('Invert', 0.0, False, (0, 0), H, W)
    with pytest.raises(NotImplementedError):
        A.op_row('Color', 1.5, False, (0, 0), H, W)
    rows = aug.rows([None, ([('Solarize', 0, 256), ('Equalize', 0, 1)], False, (3.0, 4.0))], H, W)
    assert len(rows) == 2 and rows[0][0].shape == (2, 8) and rows[0][0].dtype.is_floating_point is False
    assert rows[0][0][:, 0].tolist() == [A.IDENTITY, A.SOLARIZE] and rows[1][0][:, 0].tolist() == [A.IDENTITY, A.EQUALIZE]
    assert abs(float(rows[0][1][1, 0]) - 256 / 3) < 1e-12


def test_crop_oracle_matches_reference_fivecrop_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'crops_golden.npz'))
    from bdvcil_amd import crop_offsets
    for i in range(int(z['n'])):
        frames, crop = list(z[f'in{i}']), tuple(int(v) for v in z[f'crop{i}'])
        got = np.stack(AO.crop_frames(frames, 'FiveCrop', crop))
        assert np.array_equal(got, z[f'out{i}'])
        # TenCrop = the same crops, each followed by its mirror image
        ten = np.stack(AO.crop_frames(frames, 'TenCrop', crop))
        T = len(frames)
        for k in range(5):
            assert np.array_equal(ten[2 * k * T:(2 * k + 1) * T], got[k * T:(k + 1) * T])
            assert np.array_equal(ten[(2 * k + 1) * T:(2 * k + 2) * T], got[k * T:(k + 1) * T][:, :, ::-1])
        # the product's offset table is the oracle's
        H, W = frames[0].shape[:2]
        offs = crop_offsets('TenCrop', H, W, crop[1], crop[0])
        assert len(offs) == 10 and [o[2] for o in offs] == [0, 1] * 5
        for k, (x, y, f) in enumerate(crop_offsets('FiveCrop', H, W, crop[1], crop[0])):
            assert np.array_equal(frames[0][y:y + crop[1], x:x + crop[0]], got[k * T]) and f == 0
    assert crop_offsets('CenterCrop', 256, 340, 224, 224) == [(58, 16, 0)]
    assert crop_offsets('ThreeCrop', 256, 340, 256, 256) == [(0, 0, 0), (84, 0, 0), (42, 0, 0)]
    assert crop_offsets('TenCrop', 256, 340, 256, 256)[:4] == [(0, 0, 0), (0, 0, 1), (84, 0, 0), (84, 0, 1)]
    with pytest.raises(KeyError):
        crop_offsets('SixCrop', 10, 10, 5, 5)
    with pytest.raises(ValueError):
        crop_offsets('CenterCrop', 10, 10, 11, 5)
