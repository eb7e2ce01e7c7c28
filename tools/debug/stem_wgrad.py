"""Debug: stem weight gradient of the bf16-piece kernel on the tensors of a real training step vs the fp32-MFMA kernel and fp64."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import bdvcil_amd as bd
from bdvcil_amd import kernels as K
from oracle import tsm_oracle as O

dev = torch.device('cuda:0')
torch.manual_seed(25)
cfg = O.r50_cfg(num_classes=7, depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
mod = bd.build_model(copy.deepcopy(cfg)).to(dev)
mod.train()
gen = torch.Generator().manual_seed(125)
imgs = torch.randn(2, 8, 3, 64, 64, generator=gen).to(dev)
labels = torch.randint(0, 7, (2, 1), generator=gen).to(dev)
orig = K.conv_wgrad
cap = {}
def patched(dy, x, g, *a, **k):
    if g.Cin == 4:
        cap['dy'], cap['x'], cap['g'] = dy.clone(), x.clone(), g
    return orig(dy, x, g, *a, **k)
K.conv_wgrad = patched
import bdvcil_amd.functional as Fn
Fn.K.conv_wgrad = patched
out = mod(imgs, labels)
out['loss_cls'].backward()
torch.cuda.synchronize()
dy, x, g = cap['dy'], cap['x'], cap['g']
print('dy', dy.shape, dy.abs().max().item(), (dy == 0).float().mean().item(), 'x', x.shape, x.abs().max().item())
d3 = orig(dy, x, g, x3=True).cpu().double()
d1 = orig(dy, x, g, x3=False).cpu().double()
xr = x.cpu().double().permute(0, 3, 1, 2).contiguous()
w = torch.zeros(g.Cout, 4, g.R, g.S, dtype=torch.float64, requires_grad=True)
y = F.conv2d(xr, w, stride=g.stride, padding=g.pad)
y.backward(dy.cpu().double().permute(0, 3, 1, 2))
ref = w.grad.permute(0, 2, 3, 1)
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print('relL2 x3 vs f64', rel(d3, ref), ' f32mfma vs f64', rel(d1, ref), ' max|ref|', ref.abs().max().item())
e = (d3 - ref).abs()
print('worst entries (co, r, s, c):', [(tuple(int(v) for v in torch.unravel_index(i, e.shape)), e.flatten()[i].item(), ref.flatten()[i].item()) for i in e.flatten().topk(5).indices])
print('per-tap err norm / ref norm:', [(round(rel(d3[:, r, s], ref[:, r, s]), 6)) for r in range(g.R) for s in range(g.S)][:49])
gstep = mod.backbone.conv1.conv.weight.grad.detach().cpu().double().permute(0, 2, 3, 1)   # (Cout, R, S, 3)
print('in-step grad vs f64 (3 channels):', rel(gstep, ref[..., :3]), ' vs standalone x3:', rel(gstep, d3[..., :3]))
e = (gstep - ref[..., :3]).abs()
print('worst in-step entries:', [(tuple(int(v) for v in torch.unravel_index(i, e.shape)), e.flatten()[i].item(), ref[..., :3].flatten()[i].item()) for i in e.flatten().topk(8).indices])
print('per-co err:', [round(rel(gstep[c], ref[c, ..., :3]), 6) for c in range(0, 64, 4)])
