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
ref.train(); mod.train(); ref64.train()
rl = ref(imgs, labels); rl['loss_cls'].backward()
r64 = ref64(imgs.double(), labels); r64['loss_cls'].backward()
ol = mod(imgs.to(dev), labels.to(dev), batch_data=None); ol['loss_cls'].backward()
print('loss', rl['loss_cls'].item(), r64['loss_cls'].item(), ol['loss_cls'].item())
rp, r64p, op = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(mod.named_parameters())
for name in rp:
    if rp[name].grad is None: continue
    if 'bn.bias' in name: continue
    print(f'{name:50s} hip-vs-f32 {_rel(op[name].grad, rp[name].grad):.2e}  f32-vs-f64 {_rel(rp[name].grad.double(), r64p[name].grad):.2e}  hip-vs-f64 {_rel(op[name].grad.double().cpu(), r64p[name].grad):.2e}')
print('--- per-output-channel error concentration for the worst weights')
for name in ['backbone.layer4.1.conv1.conv.net.weight', 'backbone.layer4.1.conv2.conv.weight', 'backbone.conv1.conv.weight']:
    a, c = op[name].grad.double().cpu(), r64p[name].grad
    err = (a - c).abs().flatten(1).max(1)[0] / c.abs().max()
    top = err.topk(5)
    l2 = (a - c).norm() / c.norm()
    l2f = (rp[name].grad.double() - c).norm() / c.norm()
    print(name, 'top5 channel errs', [f'{v:.1e}' for v in top.values.tolist()], 'median', f'{err.median().item():.1e}', 'relL2 hip', f'{l2.item():.2e}', 'relL2 f32', f'{l2f.item():.2e}')
