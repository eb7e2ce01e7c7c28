"""Oracle parity of the conv kernels at BASELINE geometry: every distinct TSM-R50 conv site of SURVEY.md Appendix B at
N = 256 frames (32 clips x 8), i.e. the shapes the work planner (whole rounds / K-split / XCD remap) was tuned on.

Reference = torch CPU fp32 conv2d + oracle.temporal_shift + autograd on the same seeded tensors; dgrad runs with the
residual-gradient add and its 1-bit ReLU mask as in a block's backward.  Bars: 2e-5 of the output scale for fprop and
dgrad (reductions over <= 4608 terms, fp32 on both sides).  wgrad reduces over up to 802 816 pixels, where the fp32 CPU
reference itself is only good to ~1e-5: the whole tensor is held to 1e-4 against it and 256 sampled entries to 2e-5 against
an fp64 dot product.  Both conv arithmetics are checked against the same reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.tsm_oracle import temporal_shift

pytestmark = pytest.mark.gpu

N = 256
# (Cin, Cout, k, stride, Hin, shift)  -- SURVEY.md Appendix B
SITES = [
    (3, 64, 7, 2, 224, 0),
    (64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 0), (64, 256, 1, 1, 56, 0), (256, 64, 1, 1, 56, 1),
    (256, 128, 1, 1, 56, 1), (128, 128, 3, 2, 56, 0), (128, 512, 1, 1, 28, 0), (256, 512, 1, 2, 56, 0),
    (512, 128, 1, 1, 28, 1), (128, 128, 3, 1, 28, 0),
    (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 0), (256, 1024, 1, 1, 14, 0), (512, 1024, 1, 2, 28, 0),
    (1024, 256, 1, 1, 14, 1), (256, 256, 3, 1, 14, 0),
    (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 0), (512, 2048, 1, 1, 7, 0), (1024, 2048, 1, 2, 14, 0),
    (2048, 512, 1, 1, 7, 1), (512, 512, 3, 1, 7, 0),
]


def _err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def check_site(site, N, dev, modes=('bf16x3', 'f32mfma'), T=8):
    """fprop / dgrad (+ residual gradient and ReLU mask) / wgrad of one conv site at N frames against torch CPU fp32 (and fp64
    samples for the weight gradient), in every arithmetic of ``modes``.  Shared with tests/test_baseline_sizes_gpu.py."""
    from bdvcil_amd import kernels as K
    Cin, Cout, k, st, H, shift = site
    pad = k // 2
    gen = torch.Generator().manual_seed(1000 + Cin + 7 * Cout + k)
    x = torch.randn(N, Cin, H, H, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
    fold = Cin // 8 if shift else 0
    x.requires_grad_(True)
    w.requires_grad_(True)
    xs = temporal_shift(x, T, 8) if shift else x
    y = F.conv2d(xs, w, stride=st, padding=pad)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    y, dx_ref, dw_ref = y.detach(), x.grad, w.grad
    x, w = x.detach(), w.detach()
    stem = Cin == 3
    if stem:            # NHWC4 input, 4th channel zero; the stem needs no input gradient
        x4 = torch.zeros(N, H, H, 4)
        x4[..., :3] = x.permute(0, 2, 3, 1)
        w4 = torch.zeros(Cout, k, k, 4)
        w4[..., :3] = w.permute(0, 2, 3, 1)
        xd, wd, cin_k = x4.to(dev), w4.to(dev), 4
    else:
        xd, wd, cin_k = x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), Cin
    g = K.make_geom(N, H, H, cin_k, Cout, k, k, st, pad, T if shift else 1, fold)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    add = mask_bits = None
    if not stem:        # residual gradient + ReLU mask of the block output, as the conv1 dgrad of a block sees them
        add = torch.randn(N, H, H, Cin, generator=gen)
        m = torch.randn(N, H, H, Cin, generator=gen) > 0
        dx_ref = dx_ref + (add * m).permute(0, 3, 1, 2)
        mask_bits = torch.from_numpy(np.packbits(m.numpy().reshape(-1), bitorder='little').view(np.int32).copy()).to(dev)
        add = add.to(dev)
    # fp64 dot products for sampled weight-gradient entries
    sg = torch.Generator().manual_seed(5)
    nsmp = 256
    co = torch.randint(0, Cout, (nsmp,), generator=sg)
    ci = torch.randint(0, Cin, (nsmp,), generator=sg)
    rr = torch.randint(0, k, (nsmp,), generator=sg)
    ss = torch.randint(0, k, (nsmp,), generator=sg)
    xs_p = F.pad(xs.detach(), (pad, pad, pad, pad)).double()
    Ho = y.shape[2]
    dw_smp = torch.empty(nsmp, dtype=torch.float64)
    dyd64 = dy.double()
    for q in range(nsmp):
        win = xs_p[:, ci[q], rr[q]:rr[q] + st * (Ho - 1) + 1:st, ss[q]:ss[q] + st * (Ho - 1) + 1:st]
        dw_smp[q] = (win * dyd64[:, co[q]]).sum()
    scale_dw = dw_ref.abs().max().item()
    for mode in modes:
        prev = K.set_conv_arith(mode)
        try:
            yo = K.conv_fprop(xd, wd, g).cpu().permute(0, 3, 1, 2)
            e = _err(yo, y)
            assert e <= 2e-5, (mode, 'fprop', e)
            if not stem:
                dxo = K.conv_dgrad(dyd, wd, g, add_src=add, add_mask_src=mask_bits).cpu().permute(0, 3, 1, 2)
                e = _err(dxo, dx_ref)
                assert e <= 2e-5, (mode, 'dgrad', e)
            dwo = K.conv_wgrad(dyd, xd, g).cpu()
            dwo = dwo[..., :3] if stem else dwo
            dwo = dwo.permute(0, 3, 1, 2)
            e = _err(dwo, dw_ref)
            assert e <= 1e-4, (mode, 'wgrad vs fp32 CPU', e)
            es = (dwo[co, ci, rr, ss].double() - dw_smp).abs().max().item() / scale_dw
            assert es <= 2e-5, (mode, 'wgrad vs fp64 samples', es)
        finally:
            K.set_conv_arith('bf16x3')
            K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


@pytest.mark.parametrize('site', SITES, ids=lambda s: 'x'.join(map(str, s)))
def test_site_parity_full_size(site, dev):
    check_site(site, N, dev)
