import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdvcil_amd import kernels as K
dev = torch.device('cuda:0')
N, H, Cin, Cout, k = 256, 16, 256, 256, 3
g = K.make_geom(N, H, H, Cin, Cout, k, k, 1, k // 2)
x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
dy = torch.randn(N, g.Ho, g.Wo, Cout, device=dev)
for rep in range(3):
    K.conv_fprop(x, w, g); K.conv_dgrad(dy, w, g); K.conv_wgrad(dy, x, g)
torch.cuda.synchronize()
