"""Per-parameter gradient difference between the fp32-tensor bf16x1 step and the bf16-storage step (same weights, same batch).
Dev tool, GPU only:  python tools/debug/bf16_grad_diff.py [depth] [size] [clips]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bdvcil_amd as bd
from bdvcil_amd import kernels as K
from oracle import tsm_oracle as O

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
torch.manual_seed(5)
cfg = O.r50_cfg(num_classes=11, depth=depth, head='SimpleLinear', loss='CrossEntropyLoss', dropout_ratio=0.0)
gen = torch.Generator().manual_seed(7)
imgs, labels = torch.randn(B, 8, 3, S, S, generator=gen).to(dev), (torch.arange(B).view(B, 1) * 3 % 11).to(dev)
base = bd.build_model(copy.deepcopy(cfg)).to(dev)
res = {}
for mode in ('bf16x3', 'bf16x1', 'bf16'):
    K.set_conv_arith(mode)
    m = copy.deepcopy(base)
    m.train()
    out = m(imgs, labels)
    out['loss_cls'].backward()
    torch.cuda.synchronize()
    res[mode] = (out['loss_cls'].item(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None})
print('loss', res['bf16x3'][0], res['bf16x1'][0], res['bf16'][0])


def cmp(a, b):
    rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
    cos = (torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm() + 1e-30)).item()
    return rel, cos


print(f'{"parameter":52s} {"bf16x1 vs fp32-level":>22s} {"bf16 vs fp32-level":>22s} {"bf16 vs bf16x1":>22s}   (rel, cos)')
for n, g in res['bf16x3'][1].items():
    if '.bn.' in n:
        continue
    g1, h = res['bf16x1'][1][n], res['bf16'][1][n]
    a, b, c = cmp(g1, g), cmp(h, g), cmp(h, g1)
    print(f'{n:52s} {a[0]:10.3e} {a[1]:7.4f}    {b[0]:10.3e} {b[1]:7.4f}    {c[0]:10.3e} {c[1]:7.4f}')
