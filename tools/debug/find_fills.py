"""Debug: which Python lines launch the small fill / copy kernels of a training step (torch profiler, with stacks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bdvcil_amd as bd
from bench import model_cfg
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bd.build_model(model_cfg(50, 101, 'SimpleLinear', 'CrossEntropyLoss', 0.5)).to(dev)
model.train()
opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                     paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
engine = bd.TrainEngine(model, opt)
g = torch.Generator().manual_seed(1000)
batch = dict(imgs=torch.randn(8, 8, 3, 224, 224, generator=g).to(dev), label=torch.randint(0, 101, (8, 1), generator=g).to(dev))
for _ in range(3):
    engine.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    engine.step(batch)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::copy_', 'aten::zero_', 'aten::zeros', 'aten::_to_copy', 'aten::clone', 'aten::contiguous'):
        st = [f for f in (ev.stack or []) if 'bdvcil' in f or 'background-debiased' in f or 'bench' in f]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else '?'))] += 1
for (name, where), n in cnt.most_common(40):
    print(n, name, where)
