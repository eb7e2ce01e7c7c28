import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import tsm_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_model_gpu import _pair, _clips, _rel
dev = torch.device('cuda:0')
depth, S, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K = 11
ref, mod, cfg = _pair(depth, 'LocalSimilarityClassifier', 'LSCLoss', K=K, dev=dev)
ref64 = copy.deepcopy(ref).double()
imgs, labels = _clips(B, 8, S, K)
def tap(model, store):
    hs = []
    for name, m in model.named_modules():
        if name.startswith('backbone.layer') and name.count('.') == 2 or name in ('cls_head.avg_pool',):
            def mk(name):
                def hook(_m, _i, o):
                    store[name + ':out'] = o.detach()
                    o.register_hook(lambda g, name=name: store.__setitem__(name + ':grad', g.detach()))
                return hook
            hs.append(m.register_forward_hook(mk(name)))
    return hs
sr, s64, so = {}, {}, {}
tap(ref, sr); tap(ref64, s64); tap(mod, so)
ref.train(); mod.train(); ref64.train()
ref(imgs, labels)['loss_cls'].backward()
ref64(imgs.double(), labels)['loss_cls'].backward()
mod(imgs.to(dev), labels.to(dev), batch_data=None)['loss_cls'].backward()
for k in sorted(sr):
    a, b, c = so[k].cpu().double(), sr[k].double(), s64[k]
    print(f'{k:40s} hip-vs-f64 {_rel(a, c):.2e}  f32-vs-f64 {_rel(b, c):.2e}   max {c.abs().max().item():.3e}')
