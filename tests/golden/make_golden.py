"""Generate golden vectors for the head/loss functions from the reference's own files.

Run ONCE in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules are imported by file path (they are pure-torch); ``lsc_loss.py`` only
needs ``mmaction.models.builder.LOSSES.register_module`` as a decorator, for which a no-op
registry object is placed in ``sys.modules``.  Only inputs/outputs (data) are written to
``tests/golden/head_loss_golden.npz``; no reference source travels with the repo.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'head_loss_golden.npz')


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def main():
    mm = types.ModuleType('mmaction')
    mm_models = types.ModuleType('mmaction.models')
    mm_builder = types.ModuleType('mmaction.models.builder')
    mm_builder.LOSSES = _NoopRegistry()
    sys.modules.update({'mmaction': mm, 'mmaction.models': mm_models, 'mmaction.models.builder': mm_builder})

    cos = _load('ref_cosine_linear', 'libs/models/cil_heads/cosine_linear.py')
    inc = _load('ref_inc_net', 'libs/models/cil_heads/inc_net.py')
    lsc = _load('ref_lsc_loss', 'libs/losses/lsc_loss.py')

    out = {}
    case = 0
    for (B, D, K, P) in [(1, 64, 5, 1), (4, 512, 51, 1), (32, 256, 101, 1), (4, 128, 7, 3), (8, 2048, 26, 1)]:
        g = torch.Generator().manual_seed(100 + case)
        x = torch.randn(B, D, generator=g, requires_grad=True)
        torch.manual_seed(200 + case)
        head = cos.LSC(D, K, nb_proxies=P)
        crit = lsc.LSCLoss()
        with torch.no_grad():
            crit.eta.fill_(1.0 + 0.25 * case)
        y = torch.randint(0, K, (B,), generator=g)
        sim = head(x)
        sim.retain_grad()
        loss = crit(sim, y)
        loss.backward()
        pre = f'lsc{case}_'
        out[pre + 'x'] = x.detach().numpy()
        out[pre + 'w'] = head.weights.detach().numpy()
        out[pre + 'y'] = y.numpy()
        out[pre + 'P'] = np.int64(P)
        out[pre + 'eta'] = crit.eta.detach().numpy()
        out[pre + 'sim'] = sim.detach().numpy()
        out[pre + 'loss'] = loss.detach().numpy()
        out[pre + 'dsim'] = sim.grad.numpy()
        out[pre + 'dx'] = x.grad.numpy()
        out[pre + 'dw'] = head.weights.grad.numpy()
        out[pre + 'deta'] = crit.eta.grad.numpy()
        # update_fc keeps the old rows
        head.update_fc(K + 5)
        out[pre + 'grown_old_rows_kept'] = np.bool_(torch.equal(head.weights.detach()[:K], torch.from_numpy(out[pre + 'w'])))
        out[pre + 'grown_shape'] = np.array(head.weights.shape)
        case += 1
    out['n_lsc'] = np.int64(case)

    case = 0
    for (B, D, K) in [(1, 64, 5), (4, 512, 51), (8, 2048, 26)]:
        g = torch.Generator().manual_seed(300 + case)
        torch.manual_seed(400 + case)
        net = inc.IncrementalNet(D, K)
        with torch.no_grad():
            net.bias.normal_(generator=g)
        x = torch.randn(B, D, generator=g, requires_grad=True)
        dy = torch.randn(B, K, generator=g)
        yv = net(x)
        yv.backward(dy)
        pre = f'inc{case}_'
        out[pre + 'x'] = x.detach().numpy()
        out[pre + 'w'] = net.weight.detach().numpy()
        out[pre + 'b'] = net.bias.detach().numpy()
        out[pre + 'dy'] = dy.numpy()
        out[pre + 'out'] = yv.detach().numpy()
        out[pre + 'dx'] = x.grad.numpy()
        out[pre + 'dw'] = net.weight.grad.numpy()
        out[pre + 'db'] = net.bias.grad.numpy()
        net.update_fc(K + 3)
        out[pre + 'grown_old_rows_kept'] = np.bool_(torch.equal(net.weight.detach()[:K], torch.from_numpy(out[pre + 'w'])))
        out[pre + 'grown_b'] = net.bias.detach().numpy()
        case += 1
    out['n_inc'] = np.int64(case)

    # LSCLoss corner: hinge active (loss rows clamped to 0) and B=1
    sim = torch.tensor([[0.99, -0.5, -0.7], [0.1, 0.2, 0.9]], requires_grad=True)
    y = torch.tensor([0, 2])
    crit = lsc.LSCLoss(eta=10.0)
    loss = crit(sim, y)
    loss.backward()
    out['hinge_sim'] = sim.detach().numpy()
    out['hinge_y'] = y.numpy()
    out['hinge_loss'] = loss.detach().numpy()
    out['hinge_dsim'] = sim.grad.numpy()
    out['hinge_deta'] = crit.eta.grad.numpy()

    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
