"""Parity of the implicit-GEMM conv kernels (fprop / dgrad / wgrad, with the fused temporal shift)
against the CPU oracle: torch fp32 conv2d + oracle.temporal_shift + autograd, same seeded inputs.
Tolerance: fp32 arithmetic on both sides, only the summation order differs -> 2e-5 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.tsm_oracle import temporal_shift

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, R, stride, pad, T, fold)
CASES = [
    (8, 14, 14, 64, 128, 1, 1, 0, 1, 0),      # plain 1x1, 128-wide tile
    (16, 7, 7, 64, 64, 1, 1, 0, 8, 8),        # 1x1 + shift, fold 8 (layer1.0.conv1), 64-wide tile
    (16, 6, 6, 256, 128, 1, 1, 0, 8, 32),     # 1x1 + shift, ragged M (576 rows)
    (8, 8, 8, 64, 128, 1, 2, 0, 1, 0),        # 1x1 stride 2 (downsample)
    (4, 10, 10, 64, 64, 3, 1, 1, 1, 0),       # 3x3
    (2, 9, 9, 128, 128, 3, 1, 1, 1, 0),       # 3x3, odd size
    (8, 8, 8, 64, 128, 3, 2, 1, 8, 8),        # 3x3 stride 2 + shift (BasicBlock conv1)
    (16, 5, 5, 128, 256, 3, 2, 1, 8, 16),     # 3x3 stride 2 + shift, odd size
    (32, 7, 7, 512, 512, 3, 1, 1, 8, 64),     # R18 layer4.1.conv1 at B=4 (4 ci-blocks per tap in wgrad)
    (8, 7, 7, 256, 512, 3, 1, 1, 1, 0),       # several ci-blocks per tap, no shift
    (2, 32, 32, 4, 64, 7, 2, 3, 1, 0),        # stem 7x7 on NHWC4
    (3, 18, 22, 4, 64, 7, 2, 3, 1, 0),        # stem, non-square, ragged
    (6, 9, 9, 64, 256, 1, 1, 0, 1, 0),        # layer-1 conv3 / downsample: 256 x 64 weight-gradient tile, ragged M (486 rows)
    (16, 9, 9, 256, 64, 1, 1, 0, 8, 32),      # layer-1 conv1 (shift): 64 x 256 weight-gradient tile, ragged M
]


def _ref(x_nchw, w_oihw, stride, pad, T, fold):
    xs = x_nchw
    if fold > 0:
        xs = temporal_shift(x_nchw, T, x_nchw.shape[1] // fold)
    return F.conv2d(xs, w_oihw, stride=stride, padding=pad)


def _mk(case, seed=0):
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, R, R, generator=g) / (Cin * R * R) ** 0.5
    if Cin == 4:            # stem: 4th channel is padding
        x[:, 3] = 0
    return x, w


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-7, f'max err {err} vs scale {scale}'


@pytest.mark.parametrize('case', CASES)
def test_fprop(case, dev, conv_arith):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case)
    ref = _ref(x, w, st, pad, T, fold)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    y = K.conv_fprop(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), g)
    torch.cuda.synchronize()
    _close(y.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 64 == 0])
@pytest.mark.parametrize('with_add', [False, True])
def test_dgrad(case, with_add, dev, conv_arith):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 1)
    x.requires_grad_(True)
    y = _ref(x, w, st, pad, T, fold)
    gen = torch.Generator().manual_seed(7)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    ref = x.grad
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    add = mask = None
    if with_add:
        add = torch.randn(N, H, W, Cin, generator=gen)
        mask = torch.randn(N, H, W, Cin, generator=gen)
        ref = ref + (add * (mask > 0)).permute(0, 3, 1, 2)
    bits = None
    if mask is not None:      # the 1-bit-per-element ReLU sign mask layout written by bdv_bn_apply
        import numpy as np
        bits = torch.from_numpy(np.packbits((mask > 0).numpy().reshape(-1), bitorder='little').view(np.int32).copy()).to(dev)
    dx = K.conv_dgrad(dy.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), g,
                      add_src=None if add is None else add.to(dev), add_mask_src=bits)
    torch.cuda.synchronize()
    _close(dx.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('case', CASES)
def test_wgrad(case, dev, conv_arith):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 2)
    w.requires_grad_(True)
    y = _ref(x, w, st, pad, T, fold)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dw = K.conv_wgrad(dyd, xd, g)
    torch.cuda.synchronize()
    _close(dw.cpu(), ref)
    # accumulate (beta = 1) doubles the gradient
    dw2 = K.conv_wgrad(dyd, xd, g, dw=dw.clone(), beta=1.0)
    torch.cuda.synchronize()
    _close(dw2.cpu(), 2 * ref)


@pytest.mark.parametrize('case', [c for c in CASES if c[4] % 128 == 0 and c[3] % 32 == 0])
def test_fprop_from_bf16_pieces(case, dev):
    """bdv_conv_fprop_x3 (the default arithmetic): fp32 products formed from three bf16 pieces per operand value.  Same tolerance as
    the fp32-MFMA kernel against the CPU reference, and close to the fp32-MFMA result itself; fused statistics and the
    folded eval-mode BatchNorm go through the same epilogues."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 5)
    ref = _ref(x, w, st, pad, T, fold).permute(0, 2, 3, 1)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    y3, part3 = K.conv_fprop(xd, wd, g, bn_stats=True, x3=True)
    y1, part1 = K.conv_fprop(xd, wd, g, bn_stats=True, x3=False)
    _close(y3.cpu(), ref)
    _close(y3.cpu(), y1.cpu(), tol=4e-6)
    _close(part3.sum(1).cpu(), part1.sum(1).cpu(), tol=2e-5)
    scale, shift = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    res = torch.randn_like(y1)
    a3 = K.conv_fprop(xd, wd, g, affine=(scale, shift, res, True), x3=True)
    a1 = K.conv_fprop(xd, wd, g, affine=(scale, shift, res, True), x3=False)
    _close(a3.cpu(), a1.cpu(), tol=1e-5)
    assert torch.equal(K.conv_fprop(xd, wd, g, x3=True), y3)          # deterministic


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 128 == 0])
@pytest.mark.parametrize('with_add', [False, True])
def test_dgrad_from_bf16_pieces(case, with_add, dev):
    """bdv_conv_dgrad_x3 (the default arithmetic) against the CPU reference and the fp32-MFMA kernel, with the residual add, the
    temporal un-shift and the stride-2 parity classes going through the unchanged epilogue."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 7)
    x.requires_grad_(True)
    y = _ref(x, w, st, pad, T, fold)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(11))
    y.backward(dy)
    ref = x.grad.permute(0, 2, 3, 1)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    add = torch.randn(ref.shape, generator=torch.Generator().manual_seed(12)) if with_add else None
    d3 = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev) if with_add else None, x3=True)
    d1 = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev) if with_add else None, x3=False)
    _close(d3.cpu(), ref + add if with_add else ref)
    _close(d3.cpu(), d1.cpu(), tol=4e-6)
    assert torch.equal(K.conv_dgrad(dyd, wd, g, add_src=add.to(dev) if with_add else None, x3=True), d3)


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 64 == 0 or c[3] == 4])
def test_wgrad_from_bf16_pieces(case, dev):
    """bdv_conv_wgrad_partial_x3 (the default arithmetic) + the batched reduction against the CPU reference and the fp32-MFMA kernel."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 2)
    w.requires_grad_(True)
    y = _ref(x, w, st, pad, T, fold)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    slab, dw3 = K.conv_wgrad_partial(dyd, xd, g, x3=True)
    K.wgrad_reduce_batched([(slab, dw3)])
    dw1 = K.conv_wgrad(dyd, xd, g, x3=False)
    _close(dw3.cpu(), ref)
    _close(dw3.cpu(), dw1.cpu(), tol=4e-6)
    slab2, dw3b = K.conv_wgrad_partial(dyd, xd, g, x3=True)
    K.wgrad_reduce_batched([(slab2, dw3b)])
    assert torch.equal(dw3, dw3b)


def _wide_range(shape, lo, hi, gen):
    """Random signs, exponents uniform in [lo, hi], full 24-bit significands."""
    e = torch.randint(lo, hi + 1, shape, generator=gen).double()
    m = 1.0 + torch.rand(shape, generator=gen, dtype=torch.float64)
    sgn = torch.randint(0, 2, shape, generator=gen).double() * 2 - 1
    return (sgn * m * torch.pow(torch.tensor(2.0, dtype=torch.float64), e)).float()


@pytest.mark.parametrize('case', [(8, 7, 7, 256, 128, 1, 1, 0, 8, 32), (4, 9, 9, 128, 128, 3, 1, 1, 1, 0)])
@pytest.mark.parametrize('regime', ['2^-60..2^60', 'denormal-adjacent'])
def test_bf16_pieces_adversarial_range(case, regime, dev):
    """The six-product bf16-piece arithmetic on operands that no BatchNorm-scaled tensor would produce.

    '2^-60..2^60': activation exponents uniform in [-60, 60], weight exponents in [-60, 0] (products up to 2^60, no fp32
    overflow), random signs: the pieces hi / mid / lo of every value are normal bf16 numbers.  Bar: elementwise
    |err| <= 4e-6 * sum|a*b| against the fp64 result, the bound an fp32 FMA chain over the same K also has to meet
    (the fp32-MFMA kernels are held to the same number on the same data).

    'denormal-adjacent': activations with exponents in [-118, -108], weights near 1: the lo pieces (2^-16 relative) fall at or
    below bf16's smallest normal number 2^-126, where the hardware may flush them.  The result may then lose the lo x hi
    products, i.e. degrade to 2^-16 relative per product at worst: bar 4e-5 * sum|a*b| (an fp32 FMA chain keeps full
    precision there; tensors of this magnitude do not occur on the path: activations are BatchNorm-scaled, gradients of
    a mean loss at batch 256 are > 1e-12)."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    gen = torch.Generator().manual_seed(77)
    if regime == '2^-60..2^60':
        x = _wide_range((N, Cin, H, W), -60, 60, gen)
        w = _wide_range((Cout, Cin, R, R), -60, 0, gen)
        tol = {True: 4e-6, False: 4e-6}
    else:
        x = _wide_range((N, Cin, H, W), -118, -108, gen)
        w = _wide_range((Cout, Cin, R, R), -1, 1, gen)
        tol = {True: 4e-5, False: 4e-6}
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    xs = temporal_shift(x, T, Cin // fold) if fold > 0 else x
    ref = F.conv2d(xs.double(), w.double(), stride=st, padding=pad)
    bound = F.conv2d(xs.double().abs(), w.double().abs(), stride=st, padding=pad)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    for x3 in (True, False):
        y = K.conv_fprop(xd, wd, g, x3=x3).cpu().permute(0, 3, 1, 2).double()
        ratio = ((y - ref).abs() / (bound + 1e-300)).max().item()
        assert ratio <= tol[x3], (regime, 'fprop', 'bf16x3' if x3 else 'f32mfma', ratio)
    # dgrad and wgrad on the same operands (dy takes the activation's range)
    wide = regime == '2^-60..2^60'
    dy = _wide_range(tuple(ref.shape), -60, 40, gen) if wide else _wide_range(tuple(ref.shape), -118, -108, gen)
    if not wide:        # tiny dy against activations near 1 (tiny x tiny would underflow fp32 altogether)
        x = _wide_range((N, Cin, H, W), -1, 1, gen)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(temporal_shift(xr, T, Cin // fold) if fold > 0 else xr, wr, stride=st, padding=pad)
    yr.backward(dy.double())
    xa = x.double().abs().requires_grad_(True)
    wa = w.double().abs().requires_grad_(True)
    F.conv2d(temporal_shift(xa, T, Cin // fold) if fold > 0 else xa, wa, stride=st, padding=pad).backward(dy.double().abs())
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    for x3 in (True, False):
        dx = K.conv_dgrad(dyd, wd, g, x3=x3).cpu().permute(0, 3, 1, 2).double()
        ratio = ((dx - xr.grad).abs() / (xa.grad + 1e-300)).max().item()
        assert ratio <= tol[x3], (regime, 'dgrad', 'bf16x3' if x3 else 'f32mfma', ratio)
        dw = K.conv_wgrad(dyd, xd, g, x3=x3).cpu().permute(0, 3, 1, 2).double()
        ratio = ((dw - wr.grad).abs() / (wa.grad + 1e-300)).max().item()
        # the activation / gradient products of the wide-range regime reach 2^120: sums of N*Ho*Wo of them stay below fp32 max
        assert ratio <= tol[x3], (regime, 'wgrad', 'bf16x3' if x3 else 'f32mfma', ratio)


def _bf16_pieces(w):
    """hi / mid / lo of an fp32 tensor, round to nearest even, on the CPU."""
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


@pytest.mark.parametrize('shape', [(128, 3, 64), (256, 1, 128), (64, 3, 32)])
def test_split_weights_planes(shape, dev):
    """bdv_conv_split_weights: the three bf16 pieces of every weight, bit for bit, in the two layouts the kernels stream
    (include/bdvcil_hip.h); hi + mid + lo reproduces the fp32 value exactly."""
    import ctypes
    from bdvcil_amd import kernels as K
    from bdvcil_amd._lib import lib
    Cout, R, Cin = shape
    g = K.make_geom(8, 8, 8, Cin, Cout, R, R, 1, R // 2)
    w = torch.randn(Cout, R, R, Cin, generator=torch.Generator().manual_seed(3)) * 0.05
    w[0, 0, 0, :4] = torch.tensor([0.0, 1.0, -3.0e-39, 65504.0])
    pf, pd = K.weight_planes(w.to(dev), g)
    n = Cout * R * R * Cin
    assert pf.numel() == 6 * n == lib().bdv_conv_weight_planes_bytes(ctypes.byref(g))
    pf = pf.cpu().view(torch.bfloat16).view(3, Cin // 32, R * R, Cout, 32)
    pd = pd.cpu().view(torch.bfloat16).view(3, R * R, Cout // 32, Cin, 32)
    pieces = _bf16_pieces(w)
    for k, piece in enumerate(pieces):
        want_f = piece.view(Cout, R * R, Cin // 32, 32).permute(2, 1, 0, 3)          # [chunk][tap][co][32 ci]
        want_d = piece.view(Cout // 32, 32, R * R, Cin).permute(2, 0, 3, 1)          # [tap][co chunk][ci][32 co]
        assert torch.equal(pf[k].view(torch.int16), want_f.contiguous().view(torch.int16)), k
        assert torch.equal(pd[k].view(torch.int16), want_d.contiguous().view(torch.int16)), k
    back = pieces[0].float() + pieces[1].float() + pieces[2].float()
    normal = w.abs() > 1e-30
    assert torch.equal(back[normal], w[normal])


def test_weight_plane_cache_follows_the_weights(dev):
    """The cached planes are refreshed when the weights change: through a torch in-place op (version counter), through the
    fused SGD kernel (raw pointers: FusedSGD.step invalidates the parameters it stepped and re-splits them on the side stream),
    through ``bump_weight_epoch()`` for any other raw write, and not otherwise."""
    import bdvcil_amd as bd
    from bdvcil_amd import functional as Fn
    from bdvcil_amd import kernels as K
    g = K.make_geom(8, 8, 8, 64, 128, 1, 1, 1, 0)
    x = torch.randn(8, 8, 8, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    w0 = (torch.randn(128, 64, 1, 1, generator=torch.Generator().manual_seed(2)) * 0.1).to(dev)
    wparam = torch.nn.Parameter(w0.contiguous(memory_format=torch.channels_last))          # OIHW parameter, as in the model
    y0 = K.conv_fprop(x, Fn.weight_krsc(wparam), g, x3=True)
    assert torch.equal(K.conv_fprop(x, Fn.weight_krsc(wparam), g, x3=True), y0)
    ptr = K.weight_planes(Fn.weight_krsc(wparam), g)[0].data_ptr()
    epoch = K.WEIGHT_EPOCH
    assert K.weight_planes(Fn.weight_krsc(wparam), g)[0].data_ptr() == ptr and len([k for k in K._PLANES if k == id(wparam)]) == 1
    with torch.no_grad():
        wparam.mul_(2.0)                                                      # torch in-place op: version counter moves
    assert torch.equal(K.conv_fprop(x, Fn.weight_krsc(wparam), g, x3=True), 2 * y0)
    frozen = torch.nn.Parameter(w0.contiguous(memory_format=torch.channels_last))          # a teacher's weight: never stepped
    yf = K.conv_fprop(x, Fn.weight_krsc(frozen), g, x3=True)
    stamp_frozen = K._PLANES[id(frozen)][1]
    opt = bd.FusedSGD([wparam], lr=0.5, momentum=0.0, weight_decay=0.0)
    wparam.grad = wparam.detach().clone()                                     # w <- w - 0.5 w: back to the first values
    opt.step()                                                                # raw-pointer kernel: invalidates wparam's planes only
    assert K.WEIGHT_EPOCH == epoch and K._PLANES[id(frozen)][1] is stamp_frozen
    assert torch.equal(K.conv_fprop(x, Fn.weight_krsc(wparam), g, x3=True), y0)
    assert torch.equal(K.conv_fprop(x, Fn.weight_krsc(frozen), g, x3=True), yf)
    wparam.data.mul_(4.0)                                                     # behind torch's back: needs the global bump (x4: exact)
    K.bump_weight_epoch()
    assert K.WEIGHT_EPOCH == epoch + 1
    assert torch.equal(K.conv_fprop(x, Fn.weight_krsc(wparam), g, x3=True), 4 * y0)
    key = id(wparam)
    del wparam, opt
    import gc
    gc.collect()
    assert key not in K._PLANES                                               # the planes die with the weight


PRE_CASES = [
    (16, 14, 14, 128, 256, 1, 1, 0),      # conv3-like 1x1
    (8, 9, 9, 64, 512, 1, 1, 0),          # ragged M
    (64, 14, 14, 256, 1024, 1, 1, 0),     # K-split remainder tiles
    (8, 12, 12, 64, 64, 3, 1, 1),         # conv2 of layer 1: 3x3 halo, 64 columns
    (4, 9, 11, 128, 128, 3, 1, 1),        # 3x3, odd sizes, 128 columns (tile the planner would not pick without pre_bn)
    (8, 12, 12, 256, 256, 3, 2, 1),       # stride-2 conv2 of a first block
    (16, 7, 7, 512, 512, 3, 1, 1),
]


@pytest.mark.parametrize('case', PRE_CASES)
def test_producer_batchnorm_in_the_consumers_loaders(case, dev, monkeypatch):
    """conv_fprop / conv_wgrad with pre_bn=(scale, shift) on the RAW output of the producing conv == bdv_bn_apply followed by the
    same conv on the activation, bit for bit (the loaders use bn_apply's fused multiply-add; halo and ragged lanes stay zero);
    and the backward kernels that derive the ReLU sign from the conv output == the ones that read the 1-bit mask."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad = case
    gen = torch.Generator().manual_seed(N + Cin + R)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad)
    assert K.fprop_pre_ok(g) == (R == 1)            # the model only lets 1x1 consumers take part unless BDVCIL_PRE_BN_3X3=1
    monkeypatch.setattr(K, 'PRE_BN_1X1_ONLY', False)
    assert K.fprop_pre_ok(g)
    y_prev = torch.randn(N, H, W, Cin, generator=gen).to(dev)            # raw conv output of the producing unit
    w = (torch.randn(Cout, R, R, Cin, generator=gen) * 0.05).to(dev)
    gamma = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
    beta = (torch.randn(Cin, generator=gen) * 0.2).to(dev)
    mean, invstd, scale, shift = K.bn_train_stats(y_prev, gamma, beta, 1e-5, 0.1, None, None)
    act, mask = K.bn_apply(y_prev, scale, shift, None, True, want_mask=True)
    # forward, with the fused statistics
    ref, pref = K.conv_fprop(act, w, g, bn_stats=True)
    got, pgot = K.conv_fprop(y_prev, w, g, bn_stats=True, pre_bn=(scale, shift))
    cpu = F.conv2d(torch.relu(y_prev.cpu() * scale.cpu() + shift.cpu()).permute(0, 3, 1, 2), w.cpu().permute(0, 3, 1, 2), stride=st, padding=pad)
    _close(got.cpu().permute(0, 3, 1, 2), cpu)
    _close(got.cpu(), ref.cpu(), tol=2e-6)          # (another tile than the planner's choice may sum in another order)
    _close(pgot.double().sum(1).cpu(), pref.double().sum(1).cpu(), tol=1e-5)
    # weight gradient: the activation operand formed in the loader
    dy = torch.randn(N, g.Ho, g.Wo, Cout, generator=gen).to(dev)
    assert torch.equal(K.conv_wgrad(dy, y_prev, g, pre_bn=(scale, shift)), K.conv_wgrad(dy, act, g))
    # the gradient entering the producer's BatchNorm: statistics in the dgrad epilogue and the BatchNorm backward, sign derived from y
    if st == 1 or R > 1:
        dx_m, part_m = K.conv_dgrad(dy, w, g, bn_stats=(y_prev, mask, mean, invstd))
        dx_d, part_d = K.conv_dgrad(dy, w, g, bn_stats=(y_prev, None, mean, invstd, (scale, shift)))
        assert torch.equal(dx_m, dx_d) and torch.equal(part_m, part_d)
        a = K.bn_backward(dx_m, mask, y_prev, gamma, mean, invstd, True, stat_partial=part_m)
        b = K.bn_backward(dx_d, None, y_prev, gamma, mean, invstd, True, stat_partial=part_d, relu_affine=(scale, shift))
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    dx = K.conv_dgrad(dy, w, g)
    a = K.bn_backward(dx, mask, y_prev, gamma, mean, invstd, True)            # standalone statistics pass
    b = K.bn_backward(dx, None, y_prev, gamma, mean, invstd, True, relu_affine=(scale, shift))
    for u, v in zip(a, b):
        assert torch.equal(u, v)


PL_CASES = [
    (16, 14, 14, 128, 256, 1, 1, 0, 8, 16),     # 1x1 + shift, Cout 256
    (8, 9, 9, 128, 256, 3, 1, 1, 1, 0),         # 3x3, odd size, ragged M (648 rows)
    (8, 8, 8, 256, 512, 3, 2, 1, 8, 32),        # 3x3 stride 2 + shift
    (16, 7, 7, 256, 128, 1, 1, 0, 8, 32),       # Cout 128: 256x128 tiles
    (4, 10, 10, 128, 128, 3, 1, 1, 1, 0),
    (8, 8, 8, 256, 256, 1, 2, 0, 1, 0),         # 1x1 stride 2 (downsample)
    (8, 12, 12, 64, 64, 3, 1, 1, 1, 0),         # 64-channel layer: 256x64 tiles in both directions
    (16, 9, 9, 256, 64, 1, 1, 0, 8, 32),        # conv1 of a layer-1 bottleneck (shift), ragged M
]


@pytest.mark.parametrize('case', PL_CASES)
@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 3, 4, 5])
def test_plane_kernels_every_tile(case, tile, dev):
    """The 8-wave kernels on pre-split weight planes (bdv_conv_fprop_pl / bdv_conv_dgrad_pl), every tile configuration
    forced in turn (-1 = planner's choice): against the CPU reference (2e-5), against the fp32-MFMA kernels (4e-6), with the
    fused BatchNorm statistics, the folded eval BatchNorm, the residual / mask add and the BatchNorm-backward statistics."""
    from bdvcil_amd import kernels as K
    from bdvcil_amd._lib import check, lib
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 21)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = _ref(x, w, st, pad, T, fold)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(22))
    y.backward(dy)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    check(lib().bdv_conv_debug_force_tile(tile), 'force_tile')
    try:
        y3, part3 = K.conv_fprop(xd, wd, g, bn_stats=True, x3=True)
        y1, part1 = K.conv_fprop(xd, wd, g, bn_stats=True, x3=False)
        _close(y3.cpu(), y.detach().permute(0, 2, 3, 1))
        _close(y3, y1, tol=4e-6)
        _close(part3.sum(1), part1.sum(1), tol=2e-5)
        assert torch.equal(K.conv_fprop(xd, wd, g, x3=True), y3)
        scale, shift = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        res = torch.randn_like(y1)
        _close(K.conv_fprop(xd, wd, g, affine=(scale, shift, res, True), x3=True),
               K.conv_fprop(xd, wd, g, affine=(scale, shift, res, True), x3=False), tol=1e-5)
        gen = torch.Generator().manual_seed(23)
        add = torch.randn(N, H, W, Cin, generator=gen)
        m = torch.randn(N, H, W, Cin, generator=gen) > 0
        bits = torch.from_numpy(np.packbits(m.numpy().reshape(-1), bitorder='little').view(np.int32).copy()).to(dev)
        want = x.grad.permute(0, 2, 3, 1) + add * m
        d3 = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev), add_mask_src=bits, x3=True)
        d1 = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev), add_mask_src=bits, x3=False)
        _close(d3.cpu(), want)
        _close(d3, d1, tol=4e-6)
        dw3 = K.conv_wgrad(dyd, xd, g, x3=True)               # 8-wave weight-gradient kernel (transposed LDS reads)
        _close(dw3.cpu(), w.grad.permute(0, 2, 3, 1))
        _close(dw3, K.conv_wgrad(dyd, xd, g, x3=False), tol=4e-6)
        assert torch.equal(K.conv_wgrad(dyd, xd, g, x3=True), dw3)
        if st == 1:          # BatchNorm-backward statistics of the tensor dx feeds, taken in the epilogue
            yprev = torch.randn(N, H, W, Cin, generator=gen).to(dev)
            gamma = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
            mean, invstd, sc, sh = K.bn_train_stats(yprev, gamma, torch.zeros(Cin, device=dev), 1e-5, 0.1, None, None)
            _, mask = K.bn_apply(yprev, sc, sh, None, True, want_mask=True)
            dxs, part = K.conv_dgrad(dyd, wd, g, add_src=add.to(dev), add_mask_src=bits, bn_stats=(yprev, mask, mean, invstd), x3=True)
            assert torch.equal(dxs, d3)
            a = K.bn_backward(d3, mask, yprev, gamma, mean, invstd, True)
            b = K.bn_backward(d3, mask, yprev, gamma, mean, invstd, True, stat_partial=part)
            for u, v in zip(a, b):
                _close(u, v, tol=2e-5)
    finally:
        check(lib().bdv_conv_debug_force_tile(-1), 'force_tile')


def test_wgrad_partial_and_batched_reduce(dev):
    """Weight gradients left as split-K partial products and reduced together in one launch (what a stage's backward does)
    equal the one-call form bit for bit, for a batch of layers of very different sizes."""
    from bdvcil_amd import kernels as K
    items, want = [], []
    for ci, case in enumerate(CASES):
        N, H, W, Cin, Cout, R, st, pad, T, fold = case
        x, w = _mk(case, 2 + ci)
        g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
        dyd = torch.randn(N, g.Ho, g.Wo, Cout, generator=torch.Generator().manual_seed(40 + ci)).to(dev)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        want.append(K.conv_wgrad(dyd, xd, g))
        slab, dw = K.conv_wgrad_partial(dyd, xd, g)
        assert slab.shape[1:] == dw.shape and slab.shape[0] >= 1
        items.append((slab, dw))
    items = items * 3                                    # more than BDV_MAX_REDUCE_ITEMS = 32: several launches
    K.wgrad_reduce_batched(items)
    for (slab, dw), ref in zip(items, want * 3):
        assert torch.equal(dw, ref)
    acc = [(s, d.clone()) for s, d in items[:5]]
    K.wgrad_reduce_batched(acc, beta=1.0)
    for (s, d), ref in zip(acc, want[:5]):
        assert torch.equal(d, ref * 2)
    with pytest.raises(ValueError):
        K.wgrad_reduce_batched([(items[0][0], items[1][1])])


def test_linearity_full_size(dev, conv_arith):
    """Size-independent property at a BASELINE-size site (layer3 conv2: 256->256 3x3 on 14x14, N=256):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2), and the kernel is deterministic run to run."""
    from bdvcil_amd import kernels as K
    g = K.make_geom(256, 14, 14, 256, 256, 3, 3, 1, 1)
    gen = torch.Generator(device='cpu').manual_seed(3)
    x1 = torch.randn(256, 14, 14, 256, generator=gen).to(dev)
    x2 = torch.randn(256, 14, 14, 256, generator=gen).to(dev)
    w = (torch.randn(256, 3, 3, 256, generator=gen) / 48).to(dev)
    y1, y2 = K.conv_fprop(x1, w, g), K.conv_fprop(x2, w, g)
    y12 = K.conv_fprop(2.0 * x1 + x2, w, g)
    y1b = K.conv_fprop(x1, w, g)
    torch.cuda.synchronize()
    assert torch.equal(y1, y1b)
    err = (y12 - (2.0 * y1 + y2)).abs().max().item()
    assert err <= 2e-5 * y12.abs().max().item()


def test_bad_shapes_raise(dev):
    from bdvcil_amd import kernels as K
    from bdvcil_amd._lib import HipExtensionError
    g = K.make_geom(2, 8, 8, 64, 64, 3, 3, 1, 1)
    x = torch.zeros(2, 8, 8, 64, device=dev)
    w = torch.zeros(64, 3, 3, 64, device=dev)
    with pytest.raises(ValueError):
        K.conv_fprop(x[:, :4], w, g)                      # wrong shape
    with pytest.raises(RuntimeError):
        K.conv_fprop(x.cpu(), w.cpu(), g)                 # CPU tensors: no fallback
    bad = K.make_geom(2, 8, 8, 64, 96, 3, 3, 1, 1)        # Cout not a multiple of 64
    with pytest.raises(HipExtensionError):
        K.conv_fprop(x, torch.zeros(96, 3, 3, 64, device=dev), bad)


@pytest.mark.parametrize('shape', [(8, 14, 14, 64, 128, 1), (64, 14, 14, 256, 256, 3), (32, 28, 28, 64, 64, 3), (3, 18, 22, 4, 64, 7)])
def test_fused_bn_statistics(shape, dev, conv_arith):
    """BatchNorm batch statistics from the fprop epilogue (incl. K-split remainder tiles -> fix-up kernel) equal the
    column sums of the stored y, and bn_train_finalize equals torch's batch_norm statistics."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R = shape
    st, pad = (2, 3) if R == 7 else (1, R // 2)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad)
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(N, H, W, Cin, generator=gen) + 0.3).to(dev)
    w = (torch.randn(Cout, R, R, Cin, generator=gen) / (Cin * R * R) ** 0.5).to(dev)
    y_plain = K.conv_fprop(x, w, g)
    y, part = K.conv_fprop(x, w, g, bn_stats=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_plain)
    yc = y.reshape(-1, Cout).double().cpu()
    M = yc.shape[0]
    s1, s2 = part[0].double().sum(0).cpu(), part[1].double().sum(0).cpu()
    assert (s1 - yc.sum(0)).abs().max() <= 1e-5 * yc.abs().sum(0).max()
    assert (s2 - (yc * yc).sum(0)).abs().max() <= 1e-5 * (yc * yc).sum(0).max()
    gamma, beta = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    rmd, rvd = rm.to(dev), rv.to(dev)
    mean, invstd, scale, shift = K.bn_train_finalize(part, M, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rmd, rvd)
    ref_mean, ref_var = yc.mean(0), yc.var(0, unbiased=False)
    assert (mean.cpu().double() - ref_mean).abs().max() <= 1e-5 * (ref_mean.abs().max() + 1)
    assert ((invstd.cpu().double() - 1 / (ref_var + 1e-5).sqrt()).abs() * (ref_var + 1e-5).sqrt()).max() <= 1e-4
    assert (rmd.cpu().double() - 0.1 * ref_mean).abs().max() <= 1e-5
    assert (rvd.cpu().double() - (0.9 + 0.1 * yc.var(0, unbiased=True))).abs().max() <= 1e-4


# BatchNorm-backward statistics taken in the dgrad epilogue: dx of this conv is the gradient entering the previous
# unit's BN(+ReLU); partial[0] / partial[1] summed over the row tiles must equal sum(g) and sum(g * xhat), and
# bn_backward fed with them must give what the separate statistics pass gives.
STAT_CASES = [
    (8, 14, 14, 64, 128, 1, 1, 0, 1, 0),      # Cin = 64: 128x64 tiles
    (16, 6, 6, 256, 128, 1, 1, 0, 1, 0),      # ragged M (576 rows), 128x128 tiles
    (2, 9, 9, 128, 128, 3, 1, 1, 1, 0),       # 3x3
    (64, 14, 14, 256, 256, 3, 1, 1, 1, 0),    # 98 x 2 = 196 tiles: one partial round, K-split fix-up path
    (8, 7, 7, 256, 512, 3, 1, 1, 1, 0),
    (16, 7, 7, 256, 128, 1, 1, 0, 8, 32),     # conv1 of a bottleneck: temporal shift (pieces land in other frames) + residual
    (16, 6, 6, 64, 64, 3, 1, 1, 8, 8),        # BasicBlock conv1 with shift, 128x64 tiles
    (8, 12, 12, 128, 128, 3, 2, 1, 1, 0),     # 3x3 stride 2: one block of partial rows per parity class (4-wave kernels)
    (8, 12, 12, 256, 256, 3, 2, 1, 1, 0),     # the same on the 128x256 plane kernel
    (4, 9, 11, 256, 512, 3, 2, 1, 1, 0),      # odd sizes: the parity classes have different numbers of row tiles
]


@pytest.mark.parametrize('case', STAT_CASES)
@pytest.mark.parametrize('relu', [True, False])
def test_dgrad_fused_bn_backward_statistics(case, relu, dev, conv_arith):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    gen = torch.Generator().manual_seed(31)
    _, w = _mk(case, 6)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    dy = torch.randn(N, g.Ho, g.Wo, Cout, generator=gen).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    yprev = torch.randn(N, H, W, Cin, generator=gen).to(dev)            # conv output of the previous unit
    gamma = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
    beta = torch.zeros(Cin, device=dev)
    mean, invstd, scale, shift = K.bn_train_stats(yprev, gamma, beta, 1e-5, 0.1, None, None)
    mask = None
    if relu:
        _, mask = K.bn_apply(yprev, scale, shift, None, True, want_mask=True)
    add = torch.randn(N, H, W, Cin, generator=gen).to(dev) if fold > 0 else None      # residual path of the block
    dx_ref = K.conv_dgrad(dy, wd, g, add_src=add)
    dx, part = K.conv_dgrad(dy, wd, g, add_src=add, bn_stats=(yprev, mask, mean, invstd))
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    Mc0 = N * ((H + st - 1) // st) * ((W + st - 1) // st)        # rows of the largest input-parity class (stride 1: all rows)
    assert part.shape[0] == 2 and part.shape[2] == Cin and part.shape[1] in [st * st * ((Mc0 + bm - 1) // bm) for bm in (64, 128, 256)]
    gm = dx_ref.double().cpu()
    if relu:
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder='little').astype(bool).reshape(N, H, W, Cin)
        gm = gm * torch.from_numpy(bits)
    xhat = (yprev.double().cpu() - mean.double().cpu()) * invstd.double().cpu()
    s1 = gm.sum(dim=(0, 1, 2))
    s2 = (gm * xhat).sum(dim=(0, 1, 2))
    _close(part[0].double().sum(0).cpu(), s1, tol=1e-5)
    _close(part[1].double().sum(0).cpu(), s2, tol=1e-5)
    a = K.bn_backward(dx_ref, mask, yprev, gamma, mean, invstd, relu)
    b = K.bn_backward(dx_ref, mask, yprev, gamma, mean, invstd, relu, stat_partial=part)
    for u, v in zip(a, b):
        _close(u, v, tol=2e-5)
    if R == 1 and Cin % 64 == 0:      # a 1x1 filter with stride 2 leaves three of four input pixels unreached: no statistics
        g2 = K.make_geom(N, H, W, Cin, Cout, 1, 1, 2, 0, T, fold)
        dy2 = torch.randn(N, g2.Ho, g2.Wo, Cout, generator=gen).to(dev)
        with pytest.raises(Exception):
            K.conv_dgrad(dy2, wd, g2, bn_stats=(yprev, mask, mean, invstd))


@pytest.mark.parametrize('case', [CASES[0], CASES[2], CASES[5], CASES[8], CASES[11], (64, 14, 14, 256, 256, 3, 1, 1, 1, 0)])
@pytest.mark.parametrize('res,relu', [(False, True), (True, True), (False, False)])
def test_fprop_folded_eval_batchnorm(case, res, relu, dev, conv_arith):
    """Eval-mode BatchNorm (+ residual) (+ ReLU) folded into the fprop epilogue == conv followed by bn_apply."""
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    x, w = _mk(case, 8)
    gen = torch.Generator().manual_seed(41)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    scale = (torch.rand(Cout, generator=gen) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=gen).to(dev)
    r = torch.randn(N, g.Ho, g.Wo, Cout, generator=gen).to(dev) if res else None
    ref = K.bn_apply(K.conv_fprop(xd, wd, g), scale, shift, r, relu)
    out = K.conv_fprop(xd, wd, g, affine=(scale, shift, r, relu))
    torch.cuda.synchronize()
    _close(out, ref, tol=1e-6)
    with pytest.raises(ValueError):
        K.conv_fprop(xd, wd, g, bn_stats=True, affine=(scale, shift, r, relu))


def test_integration_md_ctypes_stub_runs(dev):
    """The hand-written ctypes binding shown in INTEGRATION.md section 2 is executed verbatim (guards doc drift) and
    its result is checked against the package's own wrapper."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    m = re.search(r"```python\nimport ctypes, torch\n(.*?)```", text, re.S)
    assert m, 'ctypes stub not found in INTEGRATION.md'
    code = 'import ctypes, torch\n' + m.group(1)
    code = code.replace("'background-debiased-video-cil_amd/csrc/libbdvcil_hip.so'",
                        repr(os.path.join(root, 'background-debiased-video-cil_amd', 'csrc', 'libbdvcil_hip.so')))
    code = code.replace('torch.randn(256, 56, 56, 256', 'torch.randn(16, 56, 56, 256').replace(
        'ConvGeom(256, 56, 56, 256', 'ConvGeom(16, 56, 56, 256').replace('torch.empty(256, 56, 56, 128', 'torch.empty(16, 56, 56, 128')
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    torch.cuda.synchronize()
    from bdvcil_amd import kernels as K
    g = K.make_geom(16, 56, 56, 256, 128, 1, 1, 1, 0, 8, 32)
    ref = K.conv_fprop(ns['x'], ns['w'], g, x3=False)        # the stub binds bdv_conv_fprop, the fp32-MFMA entry point
    assert torch.equal(ns['y'], ref)
