import torch, time
dev='cuda:0'
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/it
for mb in (256, 822, 2048):
    n=mb*1024*1024//4
    x=torch.randn(n,device=dev); y=torch.empty_like(x); z=torch.randn(n,device=dev)
    ms=t(lambda: y.copy_(x)); print(f'{mb} MB copy      : {ms:.3f} ms  {2*n*4/ms/1e6:.0f} GB/s')
    ms=t(lambda: torch.add(x,z,out=y)); print(f'{mb} MB add(2r1w) : {ms:.3f} ms  {3*n*4/ms/1e6:.0f} GB/s')
    ms=t(lambda: x.sum()); print(f'{mb} MB sum(read)  : {ms:.3f} ms  {n*4/ms/1e6:.0f} GB/s')
    ms=t(lambda: y.fill_(1.0)); print(f'{mb} MB fill(write): {ms:.3f} ms  {n*4/ms/1e6:.0f} GB/s')
