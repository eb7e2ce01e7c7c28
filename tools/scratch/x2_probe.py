"""Exploration: accuracy of set_conv_arith('bf16x2') (kernels, logits, gradients)."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
torch.set_num_threads(16)
import bdvcil_amd as bd
from bdvcil_amd import kernels as K
from oracle import tsm_oracle as O
dev = torch.device('cuda:0')

def err(a, b): return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
def pieces2(t):
    h = t.to(torch.bfloat16).float(); m = (t - h).to(torch.bfloat16).float(); return h.double(), m.double()

CASES = [(16,14,14,128,256,1,1,0), (8,9,9,128,256,3,1,1), (8,8,8,256,512,3,2,1), (16,7,7,256,128,1,1,0), (8,12,12,64,64,3,1,1), (8,28,28,512,128,1,1,0), (32,14,14,1024,256,1,1,0), (32,14,14,256,1024,1,1,0)]
for case in CASES:
    N,H,W,Cin,Cout,R,st,pad = case
    gen = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N,Cin,H,W,generator=gen); w = torch.randn(Cout,Cin,R,R,generator=gen)/(Cin*R*R)**0.5
    y64 = F.conv2d(x.double(), w.double(), stride=st, padding=pad)
    dy = torch.randn(y64.shape, generator=gen)
    xh,xm = pieces2(x); wh,wm = pieces2(w); dh,dm = pieces2(dy)
    ymod = F.conv2d(xh+xm, wh+wm, stride=st, padding=pad) - F.conv2d(xm, wm, stride=st, padding=pad)
    def dgrad(d, ww): return torch.nn.grad.conv2d_input(x.shape, ww, d, stride=st, padding=pad)
    def wgrad(d, xx): return torch.nn.grad.conv2d_weight(xx, w.shape, d, stride=st, padding=pad)
    dx64 = dgrad(dy.double(), w.double()); dxmod = dgrad(dh+dm, wh+wm) - dgrad(dm, wm)
    dw64 = wgrad(dy.double(), x.double()); dwmod = wgrad(dh+dm, xh+xm) - wgrad(dm, xm)
    g = K.make_geom(N,H,W,Cin,Cout,R,R,st,pad,1,0)
    xd = x.permute(0,2,3,1).contiguous().to(dev); wd = w.permute(0,2,3,1).contiguous().to(dev); dyd = dy.permute(0,2,3,1).contiguous().to(dev)
    out = {}
    for mode in ('bf16x3','bf16x2'):
        K.set_conv_arith(mode)
        names = [K.conv_kernel_name(g, k) for k in ('fprop','dgrad','wgrad')] if hasattr(K,'conv_kernel_name') else []
        y = K.conv_fprop(xd, wd, g).cpu().permute(0,3,1,2).double()
        dx = K.conv_dgrad(dyd, wd, g).cpu().permute(0,3,1,2).double()
        dw = K.conv_wgrad(dyd, xd, g).cpu().permute(0,3,1,2).double()
        out[mode] = (y,dx,dw)
        print(case, mode, names, 'vs fp64: %.2e %.2e %.2e' % (err(y,y64), err(dx,dx64), err(dw,dw64)), 'vs model: %.2e %.2e %.2e' % (err(y,ymod), err(dx,dxmod), err(dw,dwmod)), flush=True)
    K.set_conv_arith('bf16x3')

# logits
for depth, S in ((18, 64), (50, 224)):
    torch.manual_seed(3)
    cfg = O.r50_cfg(num_classes=11, depth=depth, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
    ref = O.build_model(copy.deepcopy(cfg)); mod = bd.build_model(copy.deepcopy(cfg)); mod.load_state_dict(ref.state_dict()); mod.to(dev)
    gen = torch.Generator().manual_seed(11)
    imgs = torch.randn(2,8,3,S,S,generator=gen)
    ref.eval(); mod.eval()
    with torch.no_grad():
        ref.test_cfg['average_clips'] = mod.test_cfg['average_clips'] = 'score'
        r = ref.forward_test(imgs)
        for mode in ('bf16x3','bf16x2','bf16x1'):
            K.set_conv_arith(mode); K.bump_weight_epoch()
            o = mod.forward_test(imgs.to(dev)).cpu()
            print('logits R%d @%d' % (depth,S), mode, 'max abs err %.3e' % (o-r).abs().max().item(), 'scale %.3f' % r.abs().max().item(), 'argmax equal', torch.equal(o.argmax(1), r.argmax(1)), flush=True)
    K.set_conv_arith('bf16x3')

# short training runs against the oracle, R18 LSC
def curves(depth, head, loss, S, B, steps, modes):
    torch.manual_seed(3)
    cfg = O.r50_cfg(num_classes=7, depth=depth, head=head, loss=loss, dropout_ratio=0.0)
    ref = O.build_model(copy.deepcopy(cfg)); init = ref.state_dict()
    gen = torch.Generator().manual_seed(105)
    imgs, labels = torch.randn(B,8,3,S,S,generator=gen), torch.randint(0,7,(B,1),generator=gen)
    ref.train(); opt_ref = O.build_sgd(ref, lr=0.01); rc = []
    g0 = None
    for i in range(steps):
        opt_ref.zero_grad(set_to_none=True); l = ref(imgs, labels)['loss_cls']; l.backward()
        if i == 0: g0 = {n: p.grad.detach().double().clone() for n,p in ref.named_parameters() if p.grad is not None}
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); opt_ref.step(); rc.append(l.item())
    print('oracle', depth, head, ['%.6f' % v for v in rc], flush=True)
    for mode in modes:
        K.set_conv_arith(mode); K.bump_weight_epoch()
        mod = bd.build_model(copy.deepcopy(cfg)); mod.load_state_dict(init); mod.to(dev); mod.train()
        out = mod(imgs.to(dev), labels.to(dev)); out['loss_cls'].backward()
        rels = sorted(((p.grad.detach().double().cpu() - g0[n]).norm() / (g0[n].norm() + 1e-300)).item() for n,p in mod.named_parameters() if p.grad is not None)
        mod.zero_grad(set_to_none=True)
        opt = bd.build_optimizer(mod, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
        eng = bd.TrainEngine(mod, opt, grad_clip=1.0)
        hc = [eng.step(dict(imgs=imgs.to(dev), label=labels.to(dev)))['loss_cls'].item() for _ in range(steps)]
        print(mode, 'step-0 grad relL2 vs fp32 oracle: median %.2e max %.2e' % (rels[len(rels)//2], rels[-1]), 'curve worst rel %.2e' % max(abs(a-b)/max(1,abs(b)) for a,b in zip(hc,rc)), ['%.6f' % v for v in hc], flush=True)
    K.set_conv_arith('bf16x3')
curves(18, 'LocalSimilarityClassifier', 'LSCLoss', 64, 4, 6, ('bf16x3','f32mfma','bf16x2','bf16x1'))
curves(50, 'SimpleLinear', 'CrossEntropyLoss', 64, 2, 6, ('bf16x3','f32mfma','bf16x2','bf16x1'))
