"""Debug: which parameter gradients differ between PRE_BN on / off, in backward order."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bdvcil_amd as bd
from bdvcil_amd import functional as Fn
from oracle import tsm_oracle as O
dev = torch.device('cuda:0')
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
cfg = O.r50_cfg(num_classes=11, depth=depth, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
mod = bd.build_model(copy.deepcopy(cfg)).to(dev)
state = copy.deepcopy(mod.state_dict())
g = torch.Generator().manual_seed(103)
imgs = torch.randn(2, 8, 3, 64, 64, generator=g).to(dev); labels = torch.randint(0, 11, (2, 1), generator=g).to(dev)
res = []
for flag in (True, False, True):
    Fn.PRE_BN = flag
    mod.load_state_dict(state); mod.zero_grad(set_to_none=True); mod.train()
    out = mod(imgs, labels, batch_data=None); out['loss_cls'].backward(); torch.cuda.synchronize()
    res.append({n: p.grad.detach().clone() for n, p in mod.named_parameters() if p.grad is not None})
names = list(res[0].keys())
print('on vs on (determinism):', [n for n in names if not torch.equal(res[0][n], res[2][n])][:5])
bad = [n for n in names if not torch.equal(res[0][n], res[1][n])]
print(len(bad), 'of', len(names), 'differ; last in forward order:', bad[-6:])
for n in bad[-6:]:
    a, b = res[0][n], res[1][n]
    print(n, tuple(a.shape), 'max rel', ((a - b).abs().max() / b.abs().max()).item())
