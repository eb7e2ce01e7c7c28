"""Generate golden vectors for ACMSmoothCE from the reference's own ``libs/losses/acm_smooth_ce.py``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_acm.py

The file only needs ``mmaction.models.builder.LOSSES.register_module`` as a decorator, for which a no-op registry object
is placed in ``sys.modules`` (as in make_golden.py).  Only inputs/outputs are written to ``tests/golden/acm_golden.npz``.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'acm_golden.npz')


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def main():
    mm = types.ModuleType('mmaction')
    mm_models = types.ModuleType('mmaction.models')
    mm_builder = types.ModuleType('mmaction.models.builder')
    mm_builder.LOSSES = _NoopRegistry()
    sys.modules.update({'mmaction': mm, 'mmaction.models': mm_models, 'mmaction.models.builder': mm_builder})
    spec = importlib.util.spec_from_file_location('ref_acm', os.path.join(REF, 'libs/losses/acm_smooth_ce.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for ci, (B, K, alpha) in enumerate([(1, 5, 4.0), (8, 51, 4.0), (32, 101, 4.0), (6, 11, 2.0)]):
        g = torch.Generator().manual_seed(600 + ci)
        score = torch.randn(B, K, generator=g, requires_grad=True)
        labels = torch.randint(0, K, (B,), generator=g)
        bg = torch.randint(-1, K, (B, 1), generator=g)
        bg[0, 0] = -1
        fg = torch.rand(B, 1, generator=g)
        fg[-1, 0] = 1.0                                           # lambda = 1: the background label has no effect
        loss = mod.ACMSmoothCE(alpha=alpha)(score, labels, {'background_label': bg.clone(), 'foreground_ratio': fg}, K)
        loss.backward()
        out.update({f'c{ci}_score': score.detach().numpy(), f'c{ci}_labels': labels.numpy(), f'c{ci}_bg': bg.numpy(),
                    f'c{ci}_fg': fg.numpy(), f'c{ci}_alpha': np.float32(alpha), f'c{ci}_loss': loss.detach().numpy(),
                    f'c{ci}_dscore': score.grad.numpy()})
    out['n'] = np.int64(4)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
