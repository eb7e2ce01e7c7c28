"""Generate golden vectors for the fixed test-time crops from the reference's own ``libs/pipelines/five_crops.py``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_crops.py

The file uses ``mmcv.is_tuple_of`` (an argument check) and ``mmaction.datasets.PIPELINES.register_module`` (a decorator);
both are provided as three-line objects in ``sys.modules`` for the import.  Written to ``tests/golden/crops_golden.npz``:
input frames and the stacked ``results['imgs']`` of ``FiveCrop`` for several frame / crop sizes.
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'crops_golden.npz')


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def main():
    mmcv = types.ModuleType('mmcv')
    mmcv.is_tuple_of = lambda seq, typ: isinstance(seq, tuple) and all(isinstance(v, typ) for v in seq)
    mm, ds = types.ModuleType('mmaction'), types.ModuleType('mmaction.datasets')
    ds.PIPELINES = _NoopRegistry()
    sys.modules.update({'mmcv': mmcv, 'mmaction': mm, 'mmaction.datasets': ds})
    spec = importlib.util.spec_from_file_location('ref_crops', os.path.join(REF, 'libs/pipelines/five_crops.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(99)
    out, n = {}, 0
    for (H, W, T, crop) in [(32, 43, 3, 32), (32, 43, 2, 24), (29, 41, 2, (20, 16)), (40, 40, 1, 33)]:
        frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
        res = mod.FiveCrop(crop)({'imgs': list(frames)})
        out[f'in{n}'] = np.stack(frames)
        out[f'out{n}'] = np.stack(res['imgs'])
        out[f'crop{n}'] = np.array(crop if isinstance(crop, tuple) else (crop, crop))
        n += 1
    out['n'] = np.int64(n)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
