"""In-process A/B of step-level switches on the bench workload (TSM-R50, 32 clips, fwd+bwd+SGD): the variants are run in
interleaved rounds in ONE process (timings from separate processes or boxes differ by more than the effects measured here).
    python tools/ab_step.py [rounds] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bdvcil_amd as bd
from bdvcil_amd import cil_step as CS
from bdvcil_amd import functional as Fn
from bdvcil_amd import kernels as K
from bench import model_cfg

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bd.build_model(model_cfg(50, 101, 'SimpleLinear', 'CrossEntropyLoss', 0.5)).to(dev)
model.train()
opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                     paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
engine = bd.TrainEngine(model, opt)
g = torch.Generator().manual_seed(1000)
batch = dict(imgs=torch.randn(32, 8, 3, 224, 224, generator=g).to(dev), label=torch.randint(0, 101, (32, 1), generator=g).to(dev))

VARIANTS = {
    'one stream': dict(side=False, ds=False, prio=False, batch=True),
    'side streams (default)': dict(side=True, ds=True, prio=False, batch=True),
    'side streams, no BatchNorm statistics across stage boundaries': dict(side=True, ds=True, prio=False, batch=True, xs=False),
    'side streams, BDVCIL_PRE_BN=1 (1x1 consumers)': dict(side=True, ds=True, prio=False, batch=True, pre=True),
}


def run(cfg, n):
    Fn.set_side_stream_enabled(cfg['side'])
    Fn.DS_SIDE = cfg['ds']
    CS._MAIN_HIGH_PRIORITY = cfg['prio']
    Fn.BATCH_WGRAD_REDUCE = cfg['batch']
    os.environ['BDVCIL_BN_NT'] = str(cfg.get('nt', 2))
    Fn.PRE_BN = cfg.get('pre', False) is True
    Fn.PRE_BN_FWD = cfg.get('pre', False) == 'fwd'
    Fn.CROSS_STAGE_STATS = cfg.get('xs', True)
    K.PRE_BN_1X1_ONLY = not cfg.get('k3', False)
    Fn.PRE_BN_WGRAD = cfg.get('wg', 'recompute')
    for _ in range(2):
        engine.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        engine.step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    engine.step(batch)
res = {k: [] for k in VARIANTS}
for r in range(rounds):
    for name, cfg in VARIANTS.items():
        res[name].append(run(cfg, steps))
for name, ts in res.items():
    print(f'{name:36s} ' + ' '.join(f'{t:7.2f}' for t in ts) + f'   median {sorted(ts)[len(ts) // 2]:7.2f} ms/step  ({32e3 / sorted(ts)[len(ts) // 2]:.1f} clips/s)')
