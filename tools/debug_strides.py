import sys; sys.path.insert(0,'.')
import torch, bench
import bdvcil_amd as bd
from bdvcil_amd import kernels as K
dev=torch.device('cuda:0')
m=bd.build_model(bench.model_cfg(50,101,'SimpleLinear','CrossEntropyLoss',0.5)).to(dev); m.train()
opt=bd.build_optimizer(m, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
x=torch.randn(1,8,3,64,64,device=dev); y=torch.randint(0,101,(1,1),device=dev)
for it in range(2):
    opt.zero_grad(set_to_none=True)
    m(x,y)['loss_cls'].backward()
    bad=[(n,tuple(p.shape),p.stride(),p.grad.stride()) for n,p in m.named_parameters() if p.grad is not None and (p.grad.stride()!=p.stride() or not K._dense_storage(p).is_contiguous())]
    print(it, len(bad)); 
    for b in bad[:6]: print('  ',b)
    opt.step()
