"""Frame pipeline on the GPU: ``bdv_resize_linear_u8`` bit-equal to oracle/resize_oracle.py (whose OpenCV arithmetic is UNPINNED: cv2 is
absent), and ``RawFrameClipLoader`` -- files -> decode -> Resize(-1, 256) -> crops -> Normalize (+ RandAugment / background mix in
training) -- against the same stages composed on the CPU from Pillow's decode, the resize oracle and numpy."""
import io
import os
import random

import numpy as np
import pytest
import torch

from oracle import resize_oracle as R
from test_jpeg_cpu import _picture

pytestmark = pytest.mark.gpu

MEAN = np.array([123.675, 116.28, 103.53], dtype=np.float32)
STD = np.array([58.395, 57.12, 57.375], dtype=np.float32)


@pytest.mark.parametrize('Hs,Ws,Hd,Wd', [(240, 320, 256, 341), (256, 341, 224, 224), (60, 80, 30, 40), (60, 80, 60, 80), (17, 23, 224, 224),
                                         (224, 224, 37, 201), (1, 1, 5, 7), (300, 200, 256, 171)])
def test_resize_equals_the_oracle(Hs, Ws, Hd, Wd, dev):
    from bdvcil_amd import kernels as K
    rng = np.random.default_rng(Hs * 1000 + Wd)
    src = rng.integers(0, 256, (3, Hs, Ws, 3)).astype(np.uint8)
    out = K.resize_linear_u8(torch.from_numpy(src).to(dev), Hd, Wd).cpu().numpy()
    for o, s in zip(out, src):
        assert np.array_equal(o, R.resize_linear_u8(s, Wd, Hd))


def test_crop_boxes_resize_equals_the_oracle(dev):
    from bdvcil_amd import kernels as K
    rng = np.random.default_rng(4)
    B, T, H, W = 5, 3, 256, 341
    src = rng.integers(0, 256, (B, T, H, W, 3)).astype(np.uint8)
    boxes = [(0, 0, 256, 256), (85, 0, 256, 224), (42, 16, 192, 224), (173, 88, 168, 168), (0, 32, 224, 224)]      # incl. same-size copy
    out = K.resize_linear_u8(torch.from_numpy(src).to(dev), 224, 224, boxes).cpu().numpy()
    assert out.shape == (B, T, 224, 224, 3)
    for b, (x, y, w, h) in enumerate(boxes):
        for t in range(T):
            assert np.array_equal(out[b, t], R.resize_linear_u8(src[b, t, y:y + h, x:x + w], 224, 224)), (b, t)
    from bdvcil_amd._lib import HipExtensionError
    with pytest.raises((HipExtensionError, RuntimeError), match='leaves the'):
        K.resize_linear_u8(torch.from_numpy(src).to(dev), 224, 224, [(200, 0, 256, 256)] * B)


@pytest.fixture(scope='module')
def rawframes(tmp_path_factory):
    """Four 'videos' of 11 - 14 JPEG frames (120 x 160, 4:2:0) + three background images of two sizes."""
    from PIL import Image
    root = tmp_path_factory.mktemp('rawframes')
    rng = np.random.default_rng(21)
    infos = []
    for v in range(4):
        d = root / f'v_{v}'
        d.mkdir()
        n = 11 + v
        base = _picture(120, 160, v % 3, rng)
        for i in range(1, n + 1):
            Image.fromarray(np.roll(base, (2 * i, 3 * i), axis=(0, 1))).save(str(d / f'img_{i:05}.jpg'), quality=85, subsampling=2)
        infos.append({'frame_dir': str(d), 'total_frames': n, 'label': v})
    bgs = []
    for k, (h, w) in enumerate(((200, 300), (200, 300), (260, 280))):
        p = root / f'bg_{k}.jpg'
        Image.fromarray(_picture(h, w, k % 3, rng)).save(str(p), quality=90)
        bgs.append(str(p))
    return infos, bgs


def _cpu_clip(info, inds, crop):
    """decode (Pillow) -> Resize(-1, 256) (oracle) -> crop (x, y, w, h) -> float32 (T, H, W, 3)."""
    from PIL import Image
    out = []
    for i in inds:
        img = np.asarray(Image.open(os.path.join(info['frame_dir'], f'img_{int(i):05}.jpg')).convert('RGB'))
        Wr, Hr = R.rescale_size(img.shape[1], img.shape[0], (-1, 256))
        img = R.resize_linear_u8(img, Wr, Hr)
        x, y, w, h = crop(Wr, Hr)
        out.append(img[y:y + h, x:x + w])
    return np.stack(out)


def test_val_and_test_pipelines(rawframes, dev):
    from bdvcil_amd.decode import RawFrameClipLoader
    infos, _ = rawframes
    loader = RawFrameClipLoader(dev, threads=4)
    for phase in ('val', 'features_extraction'):
        b = loader(infos, phase)
        assert tuple(b['imgs'].shape) == (4, 8, 3, 224, 224) and b['label'].tolist() == [[0], [1], [2], [3]]
        for k, v in enumerate(infos):
            inds = R.sample_frames(v['total_frames'], 8, test_mode=True)
            assert b['frame_inds'][k].tolist() == inds.tolist()
            want = _cpu_clip(v, inds, lambda W, H: ((W - 224) // 2, (H - 224) // 2, 224, 224)).astype(np.float32)
            want = ((want - MEAN) * (np.float32(1) / STD)).transpose(0, 3, 1, 2)
            assert np.abs(b['imgs'][k].cpu().numpy() - want).max() <= 1e-5
    b = loader(infos[:2], 'test')                                             # TenCrop(256): five crops, each followed by its flip
    assert tuple(b['imgs'].shape) == (2, 80, 3, 256, 256)
    inds = R.sample_frames(infos[0]['total_frames'], 8, test_mode=True)
    ws = (341 - 256) // 4
    first = _cpu_clip(infos[0], inds, lambda W, H: (0, 0, 256, 256)).astype(np.float32)
    centre = _cpu_clip(infos[0], inds, lambda W, H: (2 * ws, 0, 256, 256)).astype(np.float32)
    norm = lambda a: ((a - MEAN) * (np.float32(1) / STD)).transpose(0, 3, 1, 2)
    got = b['imgs'][0].cpu().numpy()
    assert np.abs(got[0:8] - norm(first)).max() <= 1e-5
    assert np.abs(got[8:16] - norm(first[:, :, ::-1])).max() <= 1e-5
    assert np.abs(got[64:72] - norm(centre)).max() <= 1e-5


def test_train_pipeline(rawframes, dev):
    from bdvcil_amd.augment import RandAugment
    from bdvcil_amd.decode import RawFrameClipLoader
    infos, bgs = rawframes
    # (1) RandAugment never fires, alpha = 0: the output is exactly decode -> Resize -> MultiScaleCrop -> Resize(224) -> Normalize
    loader = RawFrameClipLoader(dev, randAug=RandAugment(2, 10, 0.0), alpha=0.0, bg_files=bgs, threads=4)
    np.random.seed(5); random.seed(5); torch.manual_seed(5)
    b = loader(infos, 'train')
    assert tuple(b['imgs'].shape) == (4, 8, 3, 224, 224) and not b['randAug'].any()
    boxes = loader.train_front.crop_resize.last_boxes
    for k, v in enumerate(infos):
        clip = _cpu_clip(v, b['frame_inds'][k].tolist(), lambda W, H: (0, 0, W, H))
        x, y, w, h = boxes[k]
        want = np.stack([R.resize_linear_u8(f[y:y + h, x:x + w], 224, 224) for f in clip]).astype(np.float32)
        want = ((want - MEAN) * (np.float32(1) / STD)).transpose(0, 3, 1, 2)
        assert np.abs(b['imgs'][k].cpu().numpy() - want).max() <= 1e-5, k
    # (2) the configs' settings: RandAugment with probability 0.75, background mix (alpha 0.5) for the samples it skipped
    loader = RawFrameClipLoader(dev, bg_files=bgs, threads=4)
    seen_mixed = seen_aug = False
    for seed in range(6):
        np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
        b = loader(infos, 'train')
        assert tuple(b['imgs'].shape) == (4, 8, 3, 224, 224) and torch.isfinite(b['imgs']).all()
        assert b['randAug'].dtype == torch.bool and tuple(b['randAug'].shape) == (4,)
        seen_aug |= bool(b['randAug'].any())
        seen_mixed |= bool((~b['randAug']).any())
        for k in range(4):                    # frame numbers: one per segment, increasing, inside the video
            fi = b['frame_inds'][k].tolist()
            assert fi == sorted(fi) and 1 <= fi[0] and fi[-1] <= infos[k]['total_frames']
    assert seen_aug and seen_mixed
    # (3) backgrounds drawn from the dataset's own frames when no bg_files are given (comix_loader.py:134-137)
    b = RawFrameClipLoader(dev, threads=2)(infos, 'train')
    assert tuple(b['imgs'].shape) == (4, 8, 3, 224, 224)


def test_prefetch_loader_returns_the_loaders_batches(rawframes, dev):
    """One batch ahead on a worker thread and a second stream: the same batches, in order (deterministic phases bit for bit; the train
    phase under the same seeds, since the worker makes the same draws in the same order)."""
    from bdvcil_amd.decode import PrefetchLoader, RawFrameClipLoader
    infos, bgs = rawframes
    loader = RawFrameClipLoader(dev, bg_files=bgs, threads=2)
    lists = [infos[:2], infos[2:], infos[1:3], infos]
    want = [loader(l, 'val') for l in lists]
    got = list(PrefetchLoader(loader, depth=2).iterate(lists, 'val'))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert torch.equal(g['imgs'], w['imgs']) and torch.equal(g['label'], w['label']) and g['frame_dir'] == w['frame_dir']
    np.random.seed(3); random.seed(3); torch.manual_seed(3)
    want = [loader(l, 'train') for l in lists]
    np.random.seed(3); random.seed(3); torch.manual_seed(3)
    got = list(PrefetchLoader(loader, depth=3).iterate(lists, 'train'))
    for g, w in zip(got, want):
        assert torch.equal(g['randAug'], w['randAug']) and torch.equal(g['frame_inds'], w['frame_inds'])
        assert torch.equal(g['imgs'], w['imgs'])
