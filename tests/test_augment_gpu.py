"""RandAugment kernels (csrc/augment.hip) through the C ABI against the numpy oracle and the reference-generated golden
vectors: bit-exact for every operation of the table, for whole seeded RandAugment calls and at the full working size
of the pipeline (256 x 340 frames after Resize(-1, 256))."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'randaug_golden.npz')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def golden():
    z = np.load(GOLDEN)
    return z, json.loads(str(z['cases'])), json.loads(str(z['calls']))


def _run_rows(frames, rows_i, rows_d, dev):
    from bdvcil_amd import kernels as K
    oi = torch.from_numpy(np.array(rows_i, np.int64).astype(np.int32)).to(dev)
    od = torch.tensor(rows_d, dtype=torch.float64, device=dev)
    return K.randaug_apply(torch.from_numpy(frames).to(dev), oi, od).cpu().numpy()


def test_every_golden_operation_case_in_one_batch(golden, dev):
    """All 142 reference cases as one batch of 142 single-frame clips (a different operation per clip)."""
    from bdvcil_amd import augment as A
    z, cases, _ = golden
    H, W = z['imgs'].shape[1:3]
    frames = np.stack([z['imgs'][c['img']] for c in cases])[:, None]                    # (B, 1, H, W, 3)
    rows = [A.op_row(c['name'], c['val'], c['flip'], tuple(c['loc']), H, W) for c in cases]
    got = _run_rows(frames, [r[0] for r in rows], [r[1] for r in rows], dev)
    for i, c in enumerate(cases):
        assert np.array_equal(got[i, 0], z[f'op{i}']), (i, c)


def test_whole_seeded_calls_match_the_reference(golden, dev):
    from bdvcil_amd import augment as A
    z, _, calls = golden
    aug = A.RandAugment(2, 10, 0.75)
    draws, frames = [], []
    for c in calls:
        random.seed(c['seed'])
        np.random.seed(c['seed'])
        draws.append(aug.draw(24, 36))
        frames.append(np.stack([z['imgs'][k] for k in c['imgs']]))
    x = torch.from_numpy(np.stack(frames)).to(dev)                                       # (40, 2, 24, 36, 3)
    got = aug.apply_draws(x, draws).cpu().numpy()
    for b, c in enumerate(calls):
        assert (draws[b] is not None) == c['randAug']
        assert np.array_equal(got[b], z[f'call{c["seed"]}']), c
    # __call__ draws per sample in batch order from the global generators, like B consecutive reference calls
    random.seed(123)
    np.random.seed(123)
    ref_draws = [aug.draw(24, 36) for _ in range(40)]
    random.seed(123)
    np.random.seed(123)
    out, flags = aug(x)
    assert flags.tolist() == [d is not None for d in ref_draws]
    assert torch.equal(out, aug.apply_draws(x, ref_draws))


@pytest.mark.parametrize('H,W,T', [(256, 340, 8), (37, 53, 3)])
def test_full_size_batch_against_oracle(H, W, T, dev):
    """One clip per table entry (plus flipped signs), T frames each, at the pipeline's working size."""
    from bdvcil_amd import augment as A
    rng = np.random.default_rng(H * 1000 + W)
    entries = [(n, lo, hi, f) for n, lo, hi in AO.OP_TABLE for f in ((False, True) if n in ('Rotate', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY') else (False,))]
    B = len(entries)
    frames = rng.integers(0, 256, (B, T, H, W, 3), dtype=np.uint8)
    frames[B // 2:] = (frames[B // 2:].astype(np.int32) * 5 // 8 + 30).astype(np.uint8)     # limited range: autocontrast / equalize act
    frames[::3, :, : H // 2] = (frames[::3, :, : H // 2] // 16) * 16
    rows, want = [], np.empty_like(frames)
    for b, (name, lo, hi, flip) in enumerate(entries):
        val = (10.0 / 30) * float(hi - lo) + lo
        loc = (float(rng.uniform(1, W)), float(rng.uniform(1, H)))
        rows.append(A.op_row(name, val, flip, loc, H, W))
        for t in range(T):
            want[b, t] = AO.apply_op(name, frames[b, t], val, flip, loc)
    got = _run_rows(frames, [r[0] for r in rows], [r[1] for r in rows], dev)
    for b, e in enumerate(entries):
        assert np.array_equal(got[b], want[b]), (e, int((got[b] != want[b]).sum()))
    assert sum(not np.array_equal(got[b], frames[b]) for b in range(B)) >= B - 3            # the operations do act (Identity, full-range AutoContrast stay)


def test_edge_values_and_errors(dev):
    from bdvcil_amd import augment as A
    from bdvcil_amd import kernels as K
    from bdvcil_amd._lib import HipExtensionError
    rng = np.random.default_rng(3)
    H, W = 19, 23
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    const = np.full((H, W, 3), 200, np.uint8)                        # single occupied bin: autocontrast / equalize = identity
    two = const.copy()
    two[:, :3] = 10                                                   # two bins, step = 0 for equalize on a tiny image
    cases = [('AutoContrast', const, 0, False, (0, 0)), ('Equalize', const, 0, False, (0, 0)), ('Equalize', two, 0, False, (0, 0)),
             ('AutoContrast', two, 0, False, (0, 0)), ('Solarize', img, 0.0, False, (0, 0)), ('Solarize', img, 256.0, False, (0, 0)),
             ('Posterize', img, 8, False, (0, 0)), ('Posterize', img, 0.3, False, (0, 0)), ('Rotate', img, 30.0, True, (0, 0)),
             ('ShearX', img, 0.3, False, (0, 0)), ('ShearY', img, 0.3, True, (0, 0)), ('TranslateX', img, 0.45, True, (0, 0)),
             ('TranslateY', img, 0.45, False, (0, 0)), ('TranslateX', img, 0.0, False, (0, 0)), ('CutoutAbs', img, 112.0, False, (0.1, 0.1)),
             ('CutoutAbs', img, 0.0, False, (W - 0.01, H - 0.01)), ('Brightness', img, 0.05, False, (0, 0)), ('Color', img, 0.95, False, (0, 0)),
             ('Contrast', const, 0.5, False, (0, 0)), ('Sharpness', two, 0.05, False, (0, 0))]
    frames = np.stack([c[1] for c in cases])[:, None]
    rows = [A.op_row(c[0], c[2], c[3], c[4], H, W) for c in cases]
    got = _run_rows(frames, [r[0] for r in rows], [r[1] for r in rows], dev)
    for i, c in enumerate(cases):
        assert np.array_equal(got[i, 0], AO.apply_op(c[0], c[1], c[2], c[3], c[4])), (i, c[0], c[2])
    x = torch.from_numpy(frames).to(dev)
    oi = torch.zeros(len(cases), 8, dtype=torch.int32, device=dev)
    od = torch.zeros(len(cases), 4, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        K.randaug_apply(x, oi, od, out=x)                             # in place
    with pytest.raises(ValueError):
        K.randaug_apply(x[..., :2].contiguous(), oi, od)              # not RGB
    with pytest.raises(ValueError):
        K.randaug_apply(x, oi[:-1], od)                               # one row per clip
    with pytest.raises(RuntimeError):
        K.randaug_apply(x.cpu(), oi, od)                              # no CPU fallback
    with pytest.raises(HipExtensionError):
        K.randaug_apply(x[:, :, :2].contiguous(), oi, od)             # H < 3: no 3x3 neighbourhood


def test_train_front_end_randaug_then_background_mix(dev):
    """TrainClipFrontEnd = RandAugment, then background mix for exactly the samples RandAugment skipped
    (comix_loader.py:105-116), against the two oracles chained sample by sample with the same generator state."""
    import bdvcil_amd as bd
    from oracle import tsm_oracle as O
    rng = np.random.default_rng(11)
    B, T, H, W = 12, 3, 40, 56
    frames = rng.integers(0, 256, (B, T, H, W, 3), dtype=np.uint8)
    bg = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    front = bd.TrainClipFrontEnd(bd.RandAugment(2, 10, 0.75), alpha=0.5, with_randAug=True)
    random.seed(5)
    np.random.seed(5)
    imgs, rand_flags, mixed = front(torch.from_numpy(frames).to(dev), torch.from_numpy(bg).to(dev), as_nchw=True)
    random.seed(5)
    np.random.seed(5)
    want_frames, want_flags = [], []
    for b in range(B):
        fr, flag, _ = AO.rand_augment([frames[b, t] for t in range(T)], 2, 10, 0.75)
        want_frames.append(np.stack(fr))
        want_flags.append(flag)
    assert rand_flags.tolist() == want_flags and mixed.tolist() == [not f for f in want_flags]
    assert 0 < sum(want_flags) < B
    ref = O.bgmix_normalize(torch.from_numpy(np.stack(want_frames)), torch.from_numpy(bg), ~torch.tensor(want_flags), 0.5)
    assert torch.equal(imgs.cpu(), ref)
    # the NHWC4 form that feeds the stem holds the same values
    random.seed(5)
    np.random.seed(5)
    packed, _, _ = front(torch.from_numpy(frames).to(dev), torch.from_numpy(bg).to(dev))
    assert torch.equal(packed.data.cpu()[..., :3].reshape(B, T, H, W, 3).permute(0, 1, 4, 2, 3), ref)
    # without RandAugment in the pipeline the mix is a per-sample coin of probability `prob`
    plain = bd.TrainClipFrontEnd(None, alpha=0.5, prob=0.25, with_randAug=False)
    random.seed(9)
    _, rf, mx = plain(torch.from_numpy(frames).to(dev), torch.from_numpy(bg).to(dev), as_nchw=True)
    random.seed(9)
    assert mx.tolist() == [random.random() < 0.25 for _ in range(B)] and not rf.any()
    with pytest.raises(ValueError):
        bd.TrainClipFrontEnd(None, with_randAug=True)(torch.from_numpy(frames).to(dev), torch.from_numpy(bg).to(dev))


@pytest.mark.parametrize('kind,H,W,crop', [('TenCrop', 256, 340, 256), ('ThreeCrop', 256, 340, 256), ('CenterCrop', 256, 340, 224),
                                            ('FiveCrop', 29, 41, (20, 16)), ('TenCrop', 40, 40, 33)])
def test_crop_front_end(kind, H, W, crop, dev):
    """Crops (+ flips) + Normalize in one pass against numpy slicing and the oracle's normalisation, bit-exact, in both
    output layouts and in the crop-major frame order FormatShape produces."""
    import bdvcil_amd as bd
    from oracle import tsm_oracle as O
    rng = np.random.default_rng(H + W)
    B, T = 2, 8 if H > 100 else 3
    frames = rng.integers(0, 256, (B, T, H, W, 3), dtype=np.uint8)
    front = bd.CropFrontEnd(kind, crop)
    want = np.stack([np.stack(AO.crop_frames(list(frames[b]), kind, crop)) for b in range(B)])        # (B, n*T, ch, cw, 3)
    n = want.shape[1] // T
    ref = O.bgmix_normalize(torch.from_numpy(want), torch.zeros((B,) + want.shape[2:], dtype=torch.uint8),
                            torch.zeros(B, dtype=torch.bool), 0.5)                                  # (B, n*T, 3, ch, cw), no mix
    x = torch.from_numpy(frames).to(dev)
    assert torch.equal(front.as_nchw(x).cpu(), ref)
    packed = front(x)
    assert packed.batches == B and packed.num_segments == n * T and tuple(packed.shape) == tuple(ref.shape)
    assert torch.equal(packed.data.cpu()[..., :3].reshape(B, n * T, *ref.shape[-2:], 3).permute(0, 1, 4, 2, 3), ref)
    assert float(packed.data[..., 3].abs().max()) == 0.0


def test_crop_front_end_feeds_predict_step(dev):
    """TenCrop clips from the crop front-end through predict_step give the same scores / representations as the same
    crops handed over as a float (B, 10*T, 3, h, w) tensor."""
    import bdvcil_amd as bd
    from test_task_loop_gpu import _model_cfg
    torch.manual_seed(3)
    model = bd.build_model(_model_cfg(4)).to(dev).eval()
    rng = np.random.default_rng(5)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 8, 40, 52, 3), dtype=np.uint8)).to(dev)
    front = bd.CropFrontEnd('TenCrop', 32)
    pred = bd.ReprPredictor(model)
    a = pred.predict_step({'imgs': front(frames), 'label': torch.zeros(2, 1, dtype=torch.long, device=dev)})
    b = pred.predict_step({'imgs': front.as_nchw(frames), 'label': torch.zeros(2, 1, dtype=torch.long, device=dev)})
    pred.close()
    assert a['repr_'].shape == (2, 10, 512) and a['cls_score'].shape == (2, 4)
    assert torch.equal(a['cls_score'], b['cls_score']) and torch.equal(a['repr_'], b['repr_'])


def test_crop_front_end_errors(dev):
    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K
    from bdvcil_amd._lib import HipExtensionError
    x = torch.zeros(1, 2, 20, 30, 3, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        bd.CropFrontEnd('CenterCrop', 31)(x)
    with pytest.raises(KeyError):
        bd.CropFrontEnd('NoCrop', 8)
    with pytest.raises(HipExtensionError):
        K.crop_normalize_u8(x, [(25, 0, 0)], 8, 8, O_MEAN, O_STD)                  # leaves the frame
    with pytest.raises(HipExtensionError):
        K.crop_normalize_u8(x, [(0, 0, 0)] * 13, 8, 8, O_MEAN, O_STD)              # more than BDV_MAX_CROPS
    with pytest.raises(RuntimeError):
        K.crop_normalize_u8(x.cpu(), [(0, 0, 0)], 8, 8, O_MEAN, O_STD)


O_MEAN, O_STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
