"""Parity of the non-conv kernels (BN fwd/bwd, pooling, front-end, heads, losses, KD, optimizer)
against torch CPU fp32 (the oracle functions where one exists) and the golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tsm_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'head_loss_golden.npz')


def _close(a, b, tol=1e-5, atol=1e-6):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * scale + atol, f'max err {err} vs scale {scale}'


@pytest.mark.parametrize('M,C', [(1000, 64), (777, 128), (3000, 256), (130, 2048)])
@pytest.mark.parametrize('relu,res', [(True, False), (True, True), (False, False)])
def test_bn_train_fwd_bwd(M, C, relu, res, dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(M + C)
    y = (torch.randn(M, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    r = torch.randn(M, C, generator=g) if res else None
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    # reference: F.batch_norm on (M, C) treats dim 1 as channels
    o = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res:
        o = o + r
    if relu:
        o = F.relu(o)
    dout = torch.randn(M, C, generator=g)
    o.backward(dout)

    yd, gd, bd = y.detach().to(dev), gamma.detach().to(dev), beta.detach().to(dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    mean, invstd, scale, shift = K.bn_train_stats(yd, gd, bd, 1e-5, 0.1, rmd, rvd)
    mask = None
    if relu:
        out, mask = K.bn_apply(yd, scale, shift, None if r is None else r.to(dev), True, want_mask=True)
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder='little').astype(bool).reshape(M, C)
        assert np.array_equal(bits, (o.detach() > 0).numpy())            # 1 bit per element: out > 0
    else:
        out = K.bn_apply(yd, scale, shift, None if r is None else r.to(dev), False)
    _close(out, o)
    _close(rmd, rm_ref)
    _close(rvd, rv_ref)
    dy, dg, db = K.bn_backward(dout.to(dev), mask, yd, gd, mean, invstd, relu)
    _close(dy, y.grad, tol=2e-5)
    _close(dg, gamma.grad, tol=2e-5, atol=1e-4)
    _close(db, beta.grad, tol=2e-5, atol=1e-4)
    # accumulate into existing dgamma/dbeta
    _, dg2, db2 = K.bn_backward(dout.to(dev), mask, yd, gd, mean, invstd, relu, dgamma=dg.clone(), dbeta=db.clone(), beta_acc=1.0)
    _close(dg2, 2 * gamma.grad, tol=2e-5, atol=2e-4)
    # residual-path gradient helper
    if relu:
        gmask = K.relu_bwd(dout.to(dev), mask)
        _close(gmask, dout * (o > 0))


def test_bn_eval(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    C, M = 256, 500
    y = torch.randn(M, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.1
    ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training=False, eps=1e-5))
    scale, shift = K.bn_eval_params(gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    _close(K.bn_apply(y.to(dev), scale, shift, None, True), ref)


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (3, 15, 11, 64), (2, 112, 112, 64)])
def test_maxpool(shape, dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=g)
    x = F.relu(x)                      # ties at zero, as after the stem ReLU
    xc = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = F.max_pool2d(xc, 3, 2, 1)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    out, idx = K.maxpool_fwd(x.to(dev))
    _close(out.permute(0, 3, 1, 2), ref, tol=0, atol=0)
    dx = K.maxpool_bwd(dout.permute(0, 2, 3, 1).contiguous().to(dev), idx, shape)
    # gradients routed to tied zeros differ in position only where the input is 0; compare on x > 0
    # and compare the total mass per window-sum otherwise
    m = (xc.detach() > 0)
    _close(dx.permute(0, 3, 1, 2).cpu() * m, xc.grad * m)
    _close(dx.sum(), xc.grad.sum(), tol=1e-4)


def test_avgpool_and_layout(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(13)
    x = torch.randn(6, 7, 7, 128, generator=g)
    out = K.avgpool_fwd(x.to(dev))
    _close(out, x.mean(dim=(1, 2)))
    d = torch.randn(6, 128, generator=g)
    dx = K.avgpool_bwd(d.to(dev), (6, 7, 7, 128))
    _close(dx, (d / 49)[:, None, None, :].expand(6, 7, 7, 128))
    img = torch.randn(5, 3, 20, 24, generator=g)
    o4 = K.nchw3_to_nhwc4(img.to(dev)).cpu()
    assert torch.equal(o4[..., :3], img.permute(0, 2, 3, 1)) and o4[..., 3].abs().max() == 0


@pytest.mark.parametrize('alpha', [0.5, 0.3])
def test_bgmix_frontend(alpha, dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(17)
    B, T, H, W = 3, 4, 20, 28
    fr = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    bg = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    mix = torch.tensor([1, 0, 1], dtype=torch.bool)
    ref = O.bgmix_normalize(fr, bg, mix, alpha)
    o4, oc = K.bgmix_normalize_u8(fr.to(dev), bg.to(dev), mix.to(dev), alpha, O.IMG_MEAN, O.IMG_STD, True, True)
    assert torch.equal(oc.cpu(), ref)            # bit-exact for any alpha: products are rounded before the add, as on the CPU
    assert torch.equal(o4.cpu()[..., :3].reshape(B, T, H, W, 3).permute(0, 1, 4, 2, 3), oc.cpu())
    # no background at all
    o4n, _ = K.bgmix_normalize_u8(fr.to(dev), None, None, alpha, O.IMG_MEAN, O.IMG_STD, True, False)
    refn = O.bgmix_normalize(fr, bg, torch.zeros(B, dtype=torch.bool), alpha)
    assert torch.equal(o4n.cpu()[..., :3].reshape(B, T, H, W, 3).permute(0, 1, 4, 2, 3), refn)


@pytest.mark.parametrize('Hs,Ws', [(240, 320), (256, 340), (320, 240), (360, 480)])
def test_bg_resize_crop_into_the_mix(Hs, Ws, dev):
    """``Resize(256) -> RandomCrop(224)`` of the background pipeline (libs/loader/comix_loader.py:72-73) in one kernel, and its fp32
    output through the fused Normalize + blend.  Oracle: torch ``F.interpolate`` on the float image (what torchvision's Resize calls;
    PARITY UNPINNED: torchvision is not importable and its version is not pinned by the reference).  Bars: 1e-4 of a grey level for
    the resampled pixels (fp32 interpolation weights, summation order), 2e-5 of scale after the blend; offsets are drawn in
    torchvision's order from torch's global generator.  A 360 x 480 image SHRINKS: the kernel implements the filter-free form
    (torchvision < 0.17 default); the oracle is called with antialias=False there."""
    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K
    B, T = 3, 2
    gen = torch.Generator().manual_seed(Hs + Ws)
    bg = torch.randint(0, 256, (B, Hs, Ws, 3), generator=gen, dtype=torch.uint8)
    fe = bd.BackgroundCropFrontEnd(256, (224, 224))
    Hr, Wr = K.resized_size(Hs, Ws, 256)
    assert min(Hr, Wr) == 256 and (Hr, Wr) == ((256, int(256 * Ws / Hs)) if Hs <= Ws else (int(256 * Hs / Ws), 256))
    torch.manual_seed(11)
    tops, lefts = fe.draw(B, Hr, Wr)
    torch.manual_seed(11)                                   # the draws of torchvision.transforms.RandomCrop.get_params, in its order
    want = []
    for _ in range(B):
        i = int(torch.randint(0, Hr - 224 + 1, size=(1,)).item())
        j = int(torch.randint(0, Wr - 224 + 1, size=(1,)).item())
        want.append((i, j))
    assert list(zip(tops, lefts)) == want
    torch.manual_seed(11)
    out = fe(bg.to(dev))
    assert out.shape == (B, 224, 224, 3) and out.dtype == torch.float32
    ref = torch.stack([O.bg_resize_crop(bg[b], 256, 224, tops[b], lefts[b]) for b in range(B)])
    assert (out.cpu() - ref).abs().max().item() <= 1e-4
    if min(Hs, Ws) <= 256:                                  # enlarging: the antialiased form (newer torchvision default) is the same image
        ref_aa = torch.stack([O.bg_resize_crop(bg[b], 256, 224, tops[b], lefts[b], antialias=True) for b in range(B)])
        assert (ref_aa - ref).abs().max().item() <= 1e-4
    # through Normalize + blend, against the reference's formula on the oracle's crop
    fr = torch.randint(0, 256, (B, T, 224, 224, 3), generator=gen, dtype=torch.uint8)
    mix = torch.tensor([True, False, True])
    o4 = bd.BackgroundMixFrontEnd(alpha=0.5)(fr.to(dev), out, mix.to(dev)).data
    m, s_ = torch.tensor(O.IMG_MEAN), torch.tensor(O.IMG_STD)
    x = (fr.float() - m) * (1.0 / s_)
    blend = x * 0.5 + ((ref - m) / s_)[:, None] * 0.5
    want_o = torch.where(mix.view(-1, 1, 1, 1, 1), blend, x)
    got = o4.cpu()[..., :3].reshape(B, T, 224, 224, 3)
    assert (got - want_o).abs().max().item() <= 2e-5 * want_o.abs().max().item()
    with pytest.raises(ValueError):
        bd.BackgroundCropFrontEnd(200, (224, 224))(bg.to(dev))          # torchvision: crop larger than the resized image


def test_stale_weight_planes_are_caught_and_refreshed(dev, monkeypatch):
    """A conv weight written through ``.data`` does not move torch's version counter: the cached bf16 planes stay stale and the
    convolution silently runs on the OLD weights.  ``bump_weight_epoch()`` is the required call after such a write;
    BDVCIL_CHECK_PLANES=1 turns the silent case into an error."""
    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K
    g = K.make_geom(8, 14, 14, 128, 256, 1, 1, 1, 0)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(8, 14, 14, 128, generator=gen).to(dev)
    from bdvcil_amd import functional as Fn
    w = torch.nn.Parameter((torch.randn(256, 128, 1, 1, generator=gen) / 11.3).to(dev).contiguous(memory_format=torch.channels_last))
    conv = lambda: K.conv_fprop(x, Fn.weight_krsc(w), g)      # noqa: E731  (the (Cout,R,S,Cin) view the model passes: cache key = the parameter)
    assert K.conv_uses_planes(g, 'fprop')
    y0 = conv()
    w.data.mul_(2.0)                                          # torch does not see this write
    stale = conv()
    assert torch.equal(stale, y0)                             # the documented hazard: old planes, old result
    bd.bump_weight_epoch()
    fresh = conv()
    assert (fresh - 2 * y0).abs().max().item() <= 2e-5 * (2 * y0).abs().max().item()
    monkeypatch.setattr(K, 'CHECK_PLANES', True)
    bd.bump_weight_epoch()
    conv()                                                    # planes cut with a checksum
    w.data.mul_(0.5)
    with pytest.raises(RuntimeError, match='stale weight planes'):
        conv()
    bd.bump_weight_epoch([w])
    back = conv()
    assert (back - y0).abs().max().item() <= 2e-5 * y0.abs().max().item()


def test_lsc_and_loss_vs_golden(dev):
    """Golden vectors generated from the reference's own cosine_linear.py / lsc_loss.py."""
    from bdvcil_amd import kernels as K
    gz = np.load(GOLD)
    for i in range(int(gz['n_lsc'])):
        p = f'lsc{i}_'
        x, w = torch.from_numpy(gz[p + 'x']).to(dev), torch.from_numpy(gz[p + 'w']).to(dev)
        y = torch.from_numpy(gz[p + 'y']).to(dev)
        P = int(gz[p + 'P'])
        Kc = w.shape[0]
        eta = torch.from_numpy(gz[p + 'eta']).to(dev)
        sim, xn, wn, cb = K.lsc_fwd(x, w, Kc, P)
        _close(sim, torch.from_numpy(gz[p + 'sim']), tol=1e-5)
        loss, dsim, deta = K.lsc_loss(sim, y, eta, 0.6, True)
        _close(loss.reshape(()), torch.from_numpy(gz[p + 'loss']), tol=1e-5)
        _close(dsim, torch.from_numpy(gz[p + 'dsim']), tol=1e-4, atol=1e-7)
        _close(deta, torch.from_numpy(gz[p + 'deta']), tol=1e-4, atol=1e-6)
        dx, dw = K.lsc_bwd(dsim, x, w, xn, wn, cb, Kc, P)
        _close(dx, torch.from_numpy(gz[p + 'dx']), tol=1e-4, atol=1e-7)
        _close(dw, torch.from_numpy(gz[p + 'dw']), tol=1e-4, atol=1e-7)
    sim = torch.from_numpy(gz['hinge_sim']).to(dev)
    loss, dsim, deta = K.lsc_loss(sim, torch.from_numpy(gz['hinge_y']).to(dev), torch.tensor([10.0], device=dev), 0.6, True)
    _close(loss.reshape(()), torch.from_numpy(gz['hinge_loss']))
    _close(dsim, torch.from_numpy(gz['hinge_dsim']), tol=1e-4, atol=1e-7)
    _close(deta, torch.from_numpy(gz['hinge_deta']), tol=1e-4, atol=1e-7)


def test_lsc_loss_class_weights_vs_golden(dev):
    """``LSCLoss(class_weights=...)`` in both branches of libs/losses/lsc_loss.py (:50-51 the NCA form, :58 weighted cross entropy),
    incl. a negative weight behind the hinge: values and gradients from the reference's own file (make_golden_lsc_weights.py)."""
    import os
    import bdvcil_amd as bd
    gz = np.load(os.path.join(os.path.dirname(GOLD), 'lsc_weights_golden.npz'))
    for i in range(int(gz['n'])):
        p = f'c{i}_'
        nca, hinge = (bool(v) for v in gz[p + 'cfg'])
        sim = torch.from_numpy(gz[p + 'sim']).to(dev).requires_grad_(True)
        y = torch.from_numpy(gz[p + 'y']).to(dev)
        crit = bd.LSCLoss(eta=float(gz[p + 'eta']), exclude_pos_denominator=nca, hinge_proxynca=hinge,
                          class_weights=torch.from_numpy(gz[p + 'cw'])).to(dev)
        loss = crit(sim, y)
        loss.backward()
        _close(loss.detach().reshape(()), torch.from_numpy(gz[p + 'loss']), tol=1e-5)
        _close(sim.grad, torch.from_numpy(gz[p + 'dsim']), tol=1e-4, atol=1e-7)
        if nca:
            _close(crit.eta.grad, torch.from_numpy(gz[p + 'deta']), tol=1e-4, atol=1e-6)
    # the mmaction-style CrossEntropyLoss(class_weight=[...]) is the same weighted mean
    g = torch.Generator().manual_seed(9)
    score = torch.randn(6, 5, generator=g)
    lab = torch.randint(0, 5, (6,), generator=g)
    cw = [0.5, 2.0, 1.0, 0.25, 3.0]
    ref = torch.nn.functional.cross_entropy(score, lab, weight=torch.tensor(cw))
    got = bd.CrossEntropyLoss(class_weight=cw)(score.to(dev), lab.to(dev))
    _close(got.reshape(()), ref.reshape(()), tol=1e-5)


def test_linear_vs_golden(dev):
    from bdvcil_amd import kernels as K
    gz = np.load(GOLD)
    for i in range(int(gz['n_inc'])):
        p = f'inc{i}_'
        x, w, b = (torch.from_numpy(gz[p + k]).to(dev) for k in ('x', 'w', 'b'))
        dy = torch.from_numpy(gz[p + 'dy']).to(dev)
        _close(K.linear_fwd(x, w, b), torch.from_numpy(gz[p + 'out']), tol=1e-5)
        dx, dw, db = K.linear_bwd(dy, x, w)
        _close(dx, torch.from_numpy(gz[p + 'dx']), tol=1e-5)
        _close(dw, torch.from_numpy(gz[p + 'dw']), tol=1e-5)
        _close(db, torch.from_numpy(gz[p + 'db']), tol=1e-5)


def test_consensus_softce_icarl_topk(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(23)
    B, T, Kc = 6, 8, 37
    s = torch.randn(B * T, Kc, generator=g)
    cons = K.consensus_fwd(s.to(dev), B, T)
    _close(cons, s.view(B, T, Kc).mean(1))
    d = torch.randn(B, Kc, generator=g)
    _close(K.consensus_bwd(d.to(dev), T), (d / T)[:, None, :].expand(B, T, Kc).reshape(B * T, Kc))
    # plain CE
    sc = torch.randn(B, Kc, generator=g).requires_grad_(True)
    y = torch.randint(0, Kc, (B,), generator=g)
    ref = F.cross_entropy(sc, y)
    ref.backward()
    loss, dsc = K.softce_loss(sc.detach().to(dev), labels=y.to(dev))
    _close(loss.reshape(()), ref)
    _close(dsc, sc.grad, tol=1e-5, atol=1e-7)
    # iCaRL soft targets
    prev = torch.randn(B, Kc, generator=g)
    prevK = 20
    tgt_ref = O.icarl_targets(y, Kc, prev, prevK)
    bgl, fgr = torch.randint(-1, Kc, (y.numel(), 1), generator=g), torch.rand(y.numel(), 1, generator=g)
    base = K.acm_targets(y.to(dev), bgl.view(-1).clamp(min=0).to(dev), fgr.view(-1).to(dev), 4.0, Kc)
    _close(K.icarl_targets(y.to(dev), prev.to(dev), prevK, Kc, base), O.icarl_targets(y, Kc, prev, prevK, bgl, fgr), tol=1e-6)
    tgt = K.icarl_targets(y.to(dev), prev.to(dev), prevK, Kc)
    _close(tgt, tgt_ref, tol=1e-5, atol=1e-7)
    sc2 = sc.detach().clone().requires_grad_(True)
    ref2 = O.soft_target_ce(sc2, tgt_ref)
    ref2.backward()
    loss2, dsc2 = K.softce_loss(sc2.detach().to(dev), soft_targets=tgt)
    _close(loss2.reshape(()), ref2)
    _close(dsc2, sc2.grad, tol=1e-5, atol=1e-7)
    # average_clip('prob') and top-k
    sm = K.softmax_mean(s.to(dev), B, T, True)
    _close(sm, torch.softmax(s.view(B, T, Kc), 2).mean(1), tol=1e-5)
    acc = K.topk_acc(cons, y.to(dev)).cpu()
    cref = s.view(B, T, Kc).mean(1)
    assert abs(acc[0].item() - O.top_k_hits(cref, y, 1)) < 1e-6
    assert abs(acc[1].item() - O.top_k_hits(cref, y, 5)) < 1e-6


def test_dropout_statistics(dev):
    from bdvcil_amd import kernels as K
    x = torch.ones(1 << 20, device=dev)
    a = K.dropout(x, 0.5, 1234)
    b = K.dropout(x, 0.5, 1234)
    c = K.dropout(x, 0.5, 1235)
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = (a > 0).float().mean().item()
    assert abs(keep - 0.5) < 5e-3
    assert set(a.unique().tolist()) == {0.0, 2.0}
    assert torch.equal(K.dropout(x, 0.0, 1), x)


def test_kd_mse(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(29)
    a = torch.randn(4, 6, 6, 64, generator=g)
    b = torch.randn(4, 6, 6, 64, generator=g)
    ad, bdv = a.to(dev).permute(0, 3, 1, 2), b.to(dev).permute(0, 3, 1, 2)    # NCHW views over NHWC storage
    mse = K.kd_mse_fwd(ad, bdv)
    _close(mse.reshape(()), F.mse_loss(a, b))
    gs = torch.tensor([0.5], device=dev)
    d = K.kd_mse_bwd(ad, bdv, gs, 3.0)
    assert d.shape == ad.shape and d.stride() == ad.stride()
    _close(d.permute(0, 2, 3, 1), 1.5 * 2 * (a - b) / a.numel(), tol=1e-5)


def test_multi_tensor_sgd_and_clip(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(31)
    shapes = [(64, 3, 3, 64), (1,), (101,), (257, 5), (1000, 33)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    gs = [torch.randn(*s, generator=g) for s in shapes]
    lrs = [0.01, 0.05, 0.1, 0.01, 0.02]
    wds = [1e-4, 1e-4, 0.0, 1e-4, 0.0]
    ref_p = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.SGD([dict(params=[p], lr=lr, weight_decay=wd) for p, lr, wd in zip(ref_p, lrs, wds)], lr=0.01, momentum=0.9)
    dp = [p.to(dev) for p in ps]
    bufs = [torch.zeros_like(p) for p in dp]
    n = len(ps)
    numels = torch.tensor([p.numel() for p in ps], dtype=torch.int64, device=dev)
    lr_t, wd_t = torch.tensor(lrs, device=dev), torch.tensor(wds, device=dev)
    pp = torch.tensor([p.data_ptr() for p in dp], dtype=torch.int64, device=dev)
    bp = torch.tensor([p.data_ptr() for p in bufs], dtype=torch.int64, device=dev)
    sq = torch.zeros(1, device=dev)
    coef = torch.ones(1, device=dev)
    for step in range(3):
        for p, gr in zip(ref_p, gs):
            p.grad = gr.clone() * (step + 1)
        total = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        opt.step()
        dg = [(gr * (step + 1)).to(dev) for gr in gs]
        gp = torch.tensor([t.data_ptr() for t in dg], dtype=torch.int64, device=dev)
        K.multi_sqnorm(gp, numels, n, sq)
        _close(sq.sqrt().reshape(()), total.reshape(()), tol=1e-5)
        K.clip_coef(sq, 1.0, 1.0, coef)
        K.multi_sgd(pp, gp, bp, numels, lr_t, wd_t, n, 0.9, 1.0, coef)
        torch.cuda.synchronize()
        for a, b in zip(dp, ref_p):
            _close(a, b, tol=1e-5, atol=1e-7)


def test_acm_smooth_ce_matches_reference_golden(dev):
    """libs.losses.ACMSmoothCE on the HIP path vs golden vectors from the reference's own acm_smooth_ce.py."""
    import os
    import bdvcil_amd as bd
    gz = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'acm_golden.npz'))
    for i in range(int(gz['n'])):
        p = f'c{i}_'
        score = torch.from_numpy(gz[p + 'score']).to(dev).requires_grad_(True)
        K = score.shape[1]
        bg = torch.from_numpy(gz[p + 'bg']).to(dev)
        crit = bd.build_loss(dict(type='ACMSmoothCE', alpha=float(gz[p + 'alpha'])))
        loss = crit(score, torch.from_numpy(gz[p + 'labels']).to(dev),
                    {'background_label': bg, 'foreground_ratio': torch.from_numpy(gz[p + 'fg']).to(dev)}, K)
        loss.backward()
        assert abs(loss.item() - float(gz[p + 'loss'])) <= 1e-5 * max(1.0, abs(float(gz[p + 'loss'])))
        assert torch.allclose(score.grad.cpu(), torch.from_numpy(gz[p + 'dscore']), rtol=1e-4, atol=1e-7)
        assert int((bg == -1).sum()) == 0          # like the reference, the background labels are rewritten in place


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (3, 15, 11, 64), (2, 112, 112, 64), (1, 9, 14, 128)])
def test_bn_relu_maxpool_fused_equals_separate_kernels(shape, dev):
    """Stem tail in one pass == bn_apply (+ReLU, mask) followed by maxpool_fwd, bit for bit (values, arg-max codes, mask)."""
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(17)
    y = torch.randn(*shape, generator=g).to(dev)
    C = shape[-1]
    scale = (torch.rand(C, generator=g) + 0.5).to(dev)
    shift = (torch.randn(C, generator=g) * 0.5).to(dev)
    a, mask_ref = K.bn_apply(y, scale, shift, None, True, want_mask=True)
    out_ref, idx_ref = K.maxpool_fwd(a)
    out, idx, mask = K.bn_relu_maxpool_fwd(y, scale, shift)
    torch.cuda.synchronize()
    assert torch.equal(out, out_ref)
    assert torch.equal(idx, idx_ref)
    assert torch.equal(mask, mask_ref)


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (3, 15, 11, 64), (4, 112, 112, 64), (1, 9, 14, 128)])
def test_bn_backward_behind_maxpool_equals_separate_kernels(shape, dev):
    """Stem backward without the expanded pooling gradient == maxpool_bwd followed by bn_backward."""
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(19)
    y = torch.randn(*shape, generator=g).to(dev)
    C = shape[-1]
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.3).to(dev)
    mean, invstd, scale, shift = K.bn_train_stats(y, gamma, beta, 1e-5, 0.1, None, None)
    p, idx, mask = K.bn_relu_maxpool_fwd(y, scale, shift)
    dp = torch.randn(p.shape, generator=g).to(dev)
    da = K.maxpool_bwd(dp, idx, shape)
    dy_ref, dg_ref, db_ref = K.bn_backward(da, mask, y, gamma, mean, invstd, True)
    dy, dg, db = K.bn_backward_maxpool(dp, idx, mask, y, gamma, mean, invstd)
    torch.cuda.synchronize()
    _close(dy, dy_ref, tol=2e-5)
    _close(dg, dg_ref, tol=2e-5, atol=1e-4)
    _close(db, db_ref, tol=2e-5, atol=1e-4)


def test_bn_apply_with_batchnorm_on_the_residual(dev):
    """res_affine: out = relu(y*s + b + (res*rs + rb)) == applying the residual's BatchNorm in a separate pass."""
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(23)
    M, C = 1000, 256
    y, r = torch.randn(M, C, generator=g).to(dev), torch.randn(M, C, generator=g).to(dev)
    s, b, rs, rb = [torch.randn(C, generator=g).to(dev) for _ in range(4)]
    ident = K.bn_apply(r, rs, rb, None, False)
    ref, mref = K.bn_apply(y, s, b, ident, True, want_mask=True)
    out, mask = K.bn_apply(y, s, b, r, True, want_mask=True, res_affine=(rs, rb))
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(mask, mref)
    with pytest.raises(ValueError):
        K.bn_apply(y, s, b, None, True, res_affine=(rs, rb))
