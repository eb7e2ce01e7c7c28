"""Generate golden vectors for RandAugment from the reference's own ``libs/pipelines/rand_augment.py`` (+ Pillow).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_randaug.py

The file needs ``mmaction.datasets.PIPELINES.register_module`` as a class decorator only; a no-op registry object is
placed in ``sys.modules`` for the import (as in make_golden.py).  Written to ``tests/golden/randaug_golden.npz``: the
input frames, per-operation cases (operation name, magnitude, sign, cut-out centre -> output frame) and whole
``RandAugment(n=2, m=10, prob=0.75).__call__`` results for seeded ``random`` / ``np.random`` states.
"""
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np
from PIL import Image

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'randaug_golden.npz')


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def main():
    mm, ds = types.ModuleType('mmaction'), types.ModuleType('mmaction.datasets')
    ds.PIPELINES = _NoopRegistry()
    sys.modules.update({'mmaction': mm, 'mmaction.datasets': ds})
    spec = importlib.util.spec_from_file_location('ref_randaug', os.path.join(REF, 'libs/pipelines/rand_augment.py'))
    ra = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ra)

    rng = np.random.default_rng(2024)
    H, W = 24, 36
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8),                                   # full range noise
            rng.integers(40, 200, (H // 4, W // 4, 3), dtype=np.uint8).repeat(4, 0).repeat(4, 1),   # blocky, limited range
            (np.add.outer(np.arange(H) * 5, np.arange(W) * 3)[..., None] % 200 + np.array([10, 30, 55])).astype(np.uint8),
            np.full((H, W, 3), 77, np.uint8)]
    imgs[3][5:9, 7:20] = (200, 10, 90)
    out = {'imgs': np.stack(imgs)}
    cases = []

    def add(name, img_idx, val, flip=False, loc=(0.0, 0.0)):
        fn = getattr(ra, name)
        P = Image.fromarray(imgs[img_idx])
        if name == 'CutoutAbs':
            res = fn(P, val, loc)
        elif name in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
            res = fn(P, val, flip)
        else:
            res = fn(P, val)
        out[f'op{len(cases)}'] = np.array(res)
        cases.append({'name': name, 'img': img_idx, 'val': float(val), 'flip': bool(flip), 'loc': [float(loc[0]), float(loc[1])]})

    table = ra.augment_list()
    for i in range(len(imgs)):
        for fn, lo, hi in table:
            name = fn.__name__
            vals = [(10.0 / 30) * float(hi - lo) + lo]                                          # the configs' m = 10
            if i == 1:
                vals += [float(lo), float(hi), (23.0 / 30) * float(hi - lo) + lo]
            for v in vals:
                for flip in ((False, True) if name in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate') else (False,)):
                    loc = (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
                    add(name, i, v, flip, loc)
    add('CutoutAbs', 0, 9.5, False, (0.2, 0.4))
    add('CutoutAbs', 0, 9.5, False, (W - 0.5, H - 0.25))
    out['cases'] = np.array(json.dumps(cases))

    # whole calls: RandAugment(n=2, m=10, prob=0.75) as in every config, T = 2 frames per sample
    aug = ra.RandAugment(n=2, m=10, prob=0.75)
    calls = []
    for seed in range(40):
        random.seed(seed)
        np.random.seed(seed)
        frames = [imgs[seed % 3].copy(), imgs[(seed + 1) % 3].copy()]
        res = aug({'imgs': frames})
        out[f'call{seed}'] = np.stack(res['imgs'])
        calls.append({'seed': seed, 'imgs': [seed % 3, (seed + 1) % 3], 'randAug': bool(res['randAug'])})
    out['calls'] = np.array(json.dumps(calls))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', len(cases), 'op cases,', len(calls), 'calls,',
          sum(c['randAug'] for c in calls), 'augmented')


if __name__ == '__main__':
    main()
