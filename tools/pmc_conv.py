"""Launch a few conv kernels at R50 sites once each (after a warm-up) so rocprofv3 --pmc can attribute counters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K
dev = torch.device('cuda:0')
N = 256
SH = [(256, 256, 3, 1, 14, 0), (128, 512, 1, 1, 28, 0), (512, 128, 1, 1, 28, 1), (64, 64, 3, 1, 56, 0), (64, 256, 1, 1, 56, 0), (1024, 256, 1, 1, 14, 1)]
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
for (Cin, Cout, k, st, H, sh) in SH:
    g = K.make_geom(N, H, H, Cin, Cout, k, k, st, k // 2, 8, (Cin // 8) if sh else 0)
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
    for rep in range(2):
        if which in ('all', 'fprop'): K.conv_fprop(x, w, g)
        if which in ('all', 'dgrad'): K.conv_dgrad(dy, w, g)
        if which in ('all', 'wgrad'): K.conv_wgrad(dy, x, g)
    torch.cuda.synchronize()
print('done')
