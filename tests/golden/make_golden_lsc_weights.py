"""Golden vectors for ``LSCLoss(class_weights=...)`` (libs/losses/lsc_loss.py:50-51 and :58) from the reference's own file.

Run ONCE in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lsc_weights.py

``lsc_loss.py`` is imported by file path; it only needs ``mmaction.models.builder.LOSSES.register_module`` as a decorator, for
which a no-op registry object is placed in ``sys.modules`` (as in make_golden.py).  Only inputs / outputs are written
(``tests/golden/lsc_weights_golden.npz``); no reference source travels with the repo.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lsc_weights_golden.npz')


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def main():
    mm_builder = types.ModuleType('mmaction.models.builder')
    mm_builder.LOSSES = _NoopRegistry()
    sys.modules.update({'mmaction': types.ModuleType('mmaction'), 'mmaction.models': types.ModuleType('mmaction.models'),
                        'mmaction.models.builder': mm_builder})
    spec = importlib.util.spec_from_file_location('ref_lsc_loss', os.path.join(REF, 'libs/losses/lsc_loss.py'))
    lsc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lsc)

    out = {}
    case = 0
    # (B, K, exclude_pos_denominator, hinge, eta, a negative weight among them)
    for (B, K, nca, hinge, eta, neg) in [(8, 11, True, True, 1.0, False), (16, 51, True, False, 3.0, False), (8, 7, True, True, 2.0, True),
                                         (8, 11, False, True, 1.0, False), (1, 5, True, True, 1.5, False), (32, 101, False, False, 1.0, False)]:
        g = torch.Generator().manual_seed(700 + case)
        sim = (torch.rand(B, K, generator=g) * 2 - 1).requires_grad_(True)          # cosine similarities
        y = torch.randint(0, K, (B,), generator=g)
        cw = torch.rand(K, generator=g) * 2 + 0.1
        if neg:
            cw[y[0]] = -0.7                                                          # the hinge then zeroes that row's term
        crit = lsc.LSCLoss(eta=eta, exclude_pos_denominator=nca, hinge_proxynca=hinge, class_weights=cw)
        loss = crit(sim, y)
        loss.backward()
        pre = f'c{case}_'
        out[pre + 'sim'] = sim.detach().numpy()
        out[pre + 'y'] = y.numpy()
        out[pre + 'cw'] = cw.numpy()
        out[pre + 'cfg'] = np.array([int(nca), int(hinge)], dtype=np.int64)
        out[pre + 'eta'] = np.float32(eta)
        out[pre + 'loss'] = loss.detach().numpy()
        out[pre + 'dsim'] = sim.grad.numpy()
        out[pre + 'deta'] = crit.eta.grad.numpy() if crit.eta.grad is not None else np.zeros(1, dtype=np.float32)
        case += 1
    out['n'] = np.int64(case)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
