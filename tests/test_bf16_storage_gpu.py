"""bf16 STORAGE of activations and gradients (``set_conv_arith('bf16')``, ``BDV_ACT_BF16``) -- BASELINE config 5, "TSM-ResNet50
fp16 with MFMA fp16 tiles ... batch 64": the reference's ``precision=16`` counterpart.  bf16 rather than fp16: the same MFMA rate
on gfx950, fp32's exponent range, no loss scaler.  Every tensor between the stem's max-pool and the average pool is bf16 in HBM;
arithmetic, accumulators, BatchNorm statistics, weights, weight gradients and everything after the average pool stay fp32.

The reference trains in precision 32 (libs/cil/cil.py:744-756): there are no reference numerics for this mode (parity unpinned).
Stated bars:
  * element-wise kernels (BatchNorm apply / backward, pooling): the bf16-storage kernel run on bf16 tensors equals BIT FOR BIT the
    fp32-storage kernel run on the widened tensors followed by one round-to-nearest-even -- the definition of the mode;
  * conv kernels: the same against the fp32-storage ``bf16x1`` kernels (themselves tested against the CPU convolution of the
    rounded operands in test_bf16x1_gpu.py), to one bf16 ulp (2^-8 relative) plus the fp32 summation-order noise (2e-5 of
    scale: the bf16-storage launches do not K-split their remainder tiles); weight gradients (fp32 results) within 2e-5;
  * model: eval logits within 1e-1 of the fp32 CPU oracle on the small test net, arg-max allowed to differ only where the
    oracle's top-2 margin is below 1e-1; a short training run follows the oracle's loss curve within 10 % and goes down;
    a full-size TSM-R50 step: loss within 2 % of the fp32-level path's, gradients no further from it than 1.3 x the distance of
    the bf16x1 arithmetic on fp32 tensors (test_r50_step_at_full_resolution says why the bar is relative); the KD step on hooked
    bf16 stage outputs within 10 % of the oracle's terms."""
import copy

import pytest
import torch

from oracle import tsm_oracle as O

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture
def bf16_mode():
    from bdvcil_amd import kernels as K
    prev = K.set_conv_arith('bf16')
    yield
    K.set_conv_arith('bf16x3')
    K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


def _rb(shape, gen, dev, scale=1.0):
    """random values exactly representable in bf16, as (bf16 tensor, the same values in fp32)"""
    t = (torch.randn(*shape, generator=gen) * scale).to(BF).to(dev)
    return t, t.float()


def test_bn_apply_is_the_fp32_kernel_plus_one_rounding(dev):
    from bdvcil_amd import kernels as K
    gen = torch.Generator().manual_seed(1)
    M, C = 8 * 14 * 14, 128
    y16, y32 = _rb((8, 14, 14, C), gen, dev)
    r16, r32 = _rb((8, 14, 14, C), gen, dev)
    sc = (torch.rand(C, generator=gen) + 0.5).to(dev)
    sh = torch.randn(C, generator=gen).to(dev)
    rs = (torch.rand(C, generator=gen) + 0.5).to(dev)
    rb = torch.randn(C, generator=gen).to(dev)
    for res16, res32, aff, relu in ((None, None, None, True), (r16, r32, None, True), (r16, r32, (rs, rb), True), (r16, r32, None, False)):
        if relu:
            o16, m16 = K.bn_apply(y16, sc, sh, res16, True, want_mask=True, res_affine=aff)
            o32, m32 = K.bn_apply(y32, sc, sh, res32, True, want_mask=True, res_affine=aff)
            assert torch.equal(m16, m32)          # the mask is the sign of the fp32 value before the rounding
        else:
            o16 = K.bn_apply(y16, sc, sh, res16, False, res_affine=aff)
            o32 = K.bn_apply(y32, sc, sh, res32, False, res_affine=aff)
        assert o16.dtype == BF and torch.equal(o16, o32.to(BF))


@pytest.mark.parametrize('relu', [True, False])
def test_bn_backward_is_the_fp32_kernel_plus_one_rounding(relu, dev):
    from bdvcil_amd import kernels as K
    gen = torch.Generator().manual_seed(2)
    C = 256
    y16, y32 = _rb((16, 7, 7, C), gen, dev)
    d16, d32 = _rb((16, 7, 7, C), gen, dev)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    mean = y32.mean((0, 1, 2)).contiguous()
    invstd = (1.0 / (y32.var((0, 1, 2), unbiased=False) + 1e-5).sqrt()).contiguous()
    mask = None
    if relu:
        sc = gamma * invstd
        _, mask = K.bn_apply(y32, sc.contiguous(), (-mean * sc).contiguous(), None, True, want_mask=True)
    dy16, dg16, db16 = K.bn_backward(d16, mask, y16, gamma, mean, invstd, relu)
    dy32, dg32, db32 = K.bn_backward(d32, mask, y32, gamma, mean, invstd, relu)
    assert dy16.dtype == BF and torch.equal(dy16, dy32.to(BF))
    assert torch.equal(dg16, dg32) and torch.equal(db16, db32)


def test_stem_tail_and_pools_carry_the_storage_type(dev):
    from bdvcil_amd import kernels as K
    gen = torch.Generator().manual_seed(3)
    N, H, W, C = 4, 24, 24, 64
    y = torch.randn(N, H, W, C, generator=gen).to(dev)                 # the stem conv output stays fp32
    sc = (torch.rand(C, generator=gen) + 0.5).to(dev)
    sh = torch.randn(C, generator=gen).to(dev)
    p16, i16, m16 = K.bn_relu_maxpool_fwd(y, sc, sh, out_dtype=BF)
    p32, i32, m32 = K.bn_relu_maxpool_fwd(y, sc, sh)
    assert p16.dtype == BF and torch.equal(p16, p32.to(BF)) and torch.equal(i16, i32) and torch.equal(m16, m32)
    q16, j16 = K.maxpool_fwd(y, out_dtype=BF)
    q32, j32 = K.maxpool_fwd(y)
    assert torch.equal(q16, q32.to(BF)) and torch.equal(j16, j32)
    dp16, dp32 = _rb(tuple(p32.shape), gen, dev)
    assert torch.equal(K.maxpool_bwd(dp16, i32, (N, H, W, C)), K.maxpool_bwd(dp32, i32, (N, H, W, C)))
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    mean = y.mean((0, 1, 2)).contiguous()
    invstd = (1.0 / (y.var((0, 1, 2), unbiased=False) + 1e-5).sqrt()).contiguous()
    a = K.bn_backward_maxpool(dp16, i32, m32, y, gamma, mean, invstd)
    b = K.bn_backward_maxpool(dp32, i32, m32, y, gamma, mean, invstd)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and a[0].dtype == torch.float32
    x16, x32 = _rb((8, 7, 7, 512), gen, dev)
    assert torch.equal(K.avgpool_fwd(x16), K.avgpool_fwd(x32))
    dpool = torch.randn(8, 512, generator=gen).to(dev)
    g16 = K.avgpool_bwd(dpool, (8, 7, 7, 512), BF)
    assert g16.dtype == BF and torch.equal(g16, K.avgpool_bwd(dpool, (8, 7, 7, 512)).to(BF))
    a16, a32 = _rb((8, 7, 7, 512), gen, dev)
    assert torch.equal(K.add(x16, a16), K.add(x32, a32).to(BF))


# N, H, W, Cin, Cout, R, stride, pad, T, fold
CONV_CASES = [
    (16, 14, 14, 128, 256, 1, 1, 0, 8, 16),     # conv1 of a block: 1x1 + temporal shift
    (8, 9, 9, 128, 256, 3, 1, 1, 1, 0),         # ragged rows
    (8, 8, 8, 256, 512, 3, 2, 1, 1, 0),         # stride-2 3x3 (four parity classes in dgrad)
    (16, 7, 7, 256, 128, 1, 1, 0, 8, 32),
    (8, 12, 12, 64, 64, 3, 1, 1, 1, 0),         # 64-wide tiles
    (8, 8, 8, 256, 256, 1, 2, 0, 1, 0),         # downsample 1x1 stride 2
    (8, 28, 28, 64, 256, 1, 1, 0, 1, 0),        # K = 64: two K-steps
]


def _close_to_rounding(out16, ref32):
    """|out - ref| <= one bf16 ulp of ref + the fp32 summation-order noise"""
    scale = ref32.abs().max().item() + 1e-12
    err = (out16.float() - ref32).abs()
    bound = ref32.abs() * 2.0 ** -8 + 2e-5 * scale
    assert bool((err <= bound).all()), (err - bound).max().item()
    # and almost everywhere it IS the rounding of the fp32-storage result
    assert (out16 == ref32.to(BF)).float().mean().item() > 0.98


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_kernels_against_the_fp32_storage_kernels(case, dev, bf16_mode):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    gen = torch.Generator().manual_seed(sum(case))
    x16, x32 = _rb((N, H, W, Cin), gen, dev)
    w = (torch.randn(Cout, R, R, Cin, generator=gen) / (Cin * R * R) ** 0.5).to(dev)
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    g32 = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    # fprop with the fused batch statistics
    y16, part16 = K.conv_fprop(x16, w, g, bn_stats=True)
    y32, part32 = K.conv_fprop(x32, w, g32, bn_stats=True)
    assert y16.dtype == BF
    _close_to_rounding(y16, y32)
    s16, s32 = part16.sum(1), part32.sum(1)        # the statistics come from the fp32 accumulators, before the rounding
    assert (s16 - s32).abs().max().item() <= 1e-4 * (s32.abs().max().item() + 1e-12)
    # eval epilogue: folded BatchNorm + residual + ReLU
    sc = (torch.rand(Cout, generator=gen) + 0.5).to(dev)
    sh = torch.randn(Cout, generator=gen).to(dev)
    r16, r32 = _rb(tuple(y32.shape), gen, dev)
    _close_to_rounding(K.conv_fprop(x16, w, g, affine=(sc, sh, r16, True)), K.conv_fprop(x32, w, g32, affine=(sc, sh, r32, True)))
    # dgrad: plain, and with the identity-branch add + ReLU mask + fused BatchNorm-backward statistics
    dy16, dy32 = _rb(tuple(y32.shape), gen, dev)
    _close_to_rounding(K.conv_dgrad(dy16, w, g), K.conv_dgrad(dy32, w, g32))
    if st == 1 or R > 1:
        a16, a32 = _rb((N, H, W, Cin), gen, dev)
        yp16, yp32 = _rb((N, H, W, Cin), gen, dev)
        amask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N * H * W * Cin // 32,), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
        smask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N * H * W * Cin // 32,), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
        mean = yp32.mean((0, 1, 2)).contiguous()
        invstd = (1.0 / (yp32.var((0, 1, 2), unbiased=False) + 1e-5).sqrt()).contiguous()
        dx16, p16 = K.conv_dgrad(dy16, w, g, add_src=a16, add_mask_src=amask, bn_stats=(yp16, smask, mean, invstd))
        dx32, p32 = K.conv_dgrad(dy32, w, g32, add_src=a32, add_mask_src=amask, bn_stats=(yp32, smask, mean, invstd))
        _close_to_rounding(dx16, dx32)
        t16, t32 = p16.sum(1), p32.sum(1)
        assert (t16 - t32).abs().max().item() <= 1e-4 * (t32.abs().max().item() + 1e-12)
    # weight gradient: fp32 result
    dw16 = K.conv_wgrad(dy16, x16, g)
    dw32 = K.conv_wgrad(dy32, x32, g32)
    assert dw16.dtype == torch.float32
    assert (dw16 - dw32).abs().max().item() <= 2e-5 * (dw32.abs().max().item() + 1e-12)


def test_bf16_tensors_are_refused_outside_the_mode(dev):
    from bdvcil_amd import kernels as K
    g = K.make_geom(8, 8, 8, 128, 128, 1, 1, 1, 0)
    x = torch.zeros(8, 8, 8, 128, dtype=BF, device=dev)
    w = torch.zeros(128, 1, 1, 128, device=dev)
    with pytest.raises(ValueError):
        K.conv_fprop(x, w, g)          # default arithmetic (three pieces): no bf16-storage kernels


def test_model_logits_and_short_training_run(dev, bf16_mode):
    import bdvcil_amd as bd
    K_ = 7
    torch.manual_seed(3)
    cfg = O.r50_cfg(num_classes=K_, depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
    ref = O.build_model(copy.deepcopy(cfg))
    mod = bd.build_model(copy.deepcopy(cfg))
    mod.load_state_dict(ref.state_dict())
    mod.to(dev)
    gen = torch.Generator().manual_seed(105)
    imgs, labels = torch.randn(4, 8, 3, 64, 64, generator=gen), torch.randint(0, K_, (4, 1), generator=gen)
    ref.eval(); mod.eval()
    with torch.no_grad():
        ref.test_cfg['average_clips'] = mod.test_cfg['average_clips'] = 'score'
        r = ref.forward_test(imgs)
        o = mod.forward_test(imgs.to(dev)).cpu()
    assert o.dtype == torch.float32
    assert (o - r).abs().max().item() <= 1e-1, (o - r).abs().max().item()
    top2 = r.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 1e-1
    assert torch.equal(o.argmax(1)[sure], r.argmax(1)[sure])
    ref.train(); mod.train()
    opt_ref = O.build_sgd(ref, lr=0.01)
    opt = bd.build_optimizer(mod, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                       paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
    engine = bd.TrainEngine(mod, opt, grad_clip=1.0)
    batch = dict(imgs=imgs.to(dev), label=labels.to(dev))
    ref_curve, hip_curve = [], []
    for _ in range(6):
        opt_ref.zero_grad(set_to_none=True)
        loss = ref(imgs, labels)['loss_cls']
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt_ref.step()
        ref_curve.append(loss.item())
        hip_curve.append(engine.step(batch)['loss_cls'].item())
    assert hip_curve[-1] < hip_curve[0]
    for a, b in zip(hip_curve, ref_curve):
        assert abs(a - b) <= 1e-1 * max(1.0, abs(b)), (hip_curve, ref_curve)
    for p in mod.parameters():
        assert p.dtype == torch.float32 and (p.grad is None or p.grad.dtype == torch.float32)


def test_r50_step_at_full_resolution(dev):
    """TSM-R50, 224 x 224, four clips, one forward + backward in three arithmetics on the same weights and batch: the default
    fp32-level one, ``bf16x1`` on fp32 tensors and ``bf16`` storage.

    A random-init network on noise clips is a poor instrument for gradients in ANY reduced precision: behind the average pool
    and the consensus mean the incoming gradient is constant over a clip, each BatchNorm backward projects the batch-constant and
    xhat-aligned parts out, and what is left amplifies every rounding (measured here: the bf16x1 arithmetic on fp32 tensors is
    already 0.73 relative / cosine 0.74 away from the fp32-level gradient at the LAST conv, 1.2 / 0.29 at the stem, with 4 or 16
    clips alike).  So the gradient bar is relative: bf16 storage may be at most 1.3 x as far from the fp32-level gradient as the
    bf16x1 arithmetic on fp32 tensors already is (measured 1.08 - 1.14), tensor by tensor; the classifier gradients, which see
    only the forward, within 8 %; the loss within 2 %.  Kernel-level exactness is the job of the tests above."""
    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K
    torch.manual_seed(5)
    cfg = O.r50_cfg(num_classes=11, depth=50, head='SimpleLinear', loss='CrossEntropyLoss', dropout_ratio=0.0)
    gen = torch.Generator().manual_seed(7)
    imgs, labels = torch.randn(4, 8, 3, 224, 224, generator=gen).to(dev), torch.tensor([[6], [2], [9], [0]]).to(dev)
    base = bd.build_model(copy.deepcopy(cfg)).to(dev)
    losses, grads = {}, {}
    for mode in ('bf16x3', 'bf16x1', 'bf16'):
        prev = K.set_conv_arith(mode)
        try:
            m = copy.deepcopy(base)
            m.train()
            out = m(imgs, labels)
            out['loss_cls'].backward()
            torch.cuda.synchronize()
            losses[mode] = out['loss_cls'].item()
            grads[mode] = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            K.set_conv_arith('bf16x3')
            K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev
    assert grads['bf16'].keys() == grads['bf16x3'].keys()
    assert all(torch.isfinite(g).all() for g in grads['bf16'].values())
    assert abs(losses['bf16'] - losses['bf16x3']) <= 2e-2 * abs(losses['bf16x3']), losses

    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-30)).item()
    worst = 0.0
    for n, g in grads['bf16x3'].items():
        if n.startswith('cls_head'):
            assert rel(grads['bf16'][n], g) <= 8e-2, (n, rel(grads['bf16'][n], g))
        elif n.endswith('conv.weight') or n.endswith('net.weight'):
            ratio = rel(grads['bf16'][n], g) / max(rel(grads['bf16x1'][n], g), 1e-3)
            worst = max(worst, ratio)
            assert ratio <= 1.3, (n, ratio)
    print(f'bf16 storage / bf16x1 distance to the fp32-level gradient, worst conv weight: {worst:.3f}')


def _structured_clips(B, K_, S, gen):
    """Non-noise clips: every class is a smooth pattern (two spatial frequencies, an orientation and a drift over the eight frames
    that depend on the label) on all three channels, plus 10 % noise -- something a network can fit, with distinct labels per clip."""
    labels = torch.randperm(K_, generator=gen)[:B]
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing='ij')
    clips = torch.empty(B, 8, 3, S, S)
    for b in range(B):
        k = float(labels[b])
        th = 0.37 * k
        u = xx * torch.cos(torch.tensor(th)) + yy * torch.sin(torch.tensor(th))
        for t in range(8):
            ph = 0.3 * t * (1 + k % 3)
            base = torch.sin((2 + k % 5) * 3.1416 * u + ph) + 0.5 * torch.cos((1 + k % 4) * 3.1416 * (xx - yy) - ph)
            clips[b, t] = torch.stack([base, base.roll(int(k) + 1, 0), base.roll(int(k) + 1, 1)])
    clips += 0.1 * torch.randn(clips.shape, generator=gen)
    return clips, labels.view(B, 1)


def test_r50_gradient_fidelity_on_a_conditioned_problem(dev):
    """A gradient instrument for BASELINE config 5 that can fail: TSM-R50 at 224 x 224 on eight STRUCTURED clips with eight
    distinct labels (not noise), three parts:

    (1) the optimisation itself: 40 SGD steps from the same initial weights in the fp32-level arithmetic and in ``bf16`` storage --
        the reduced-precision run must fit the batch as the fp32-level run does (loss below 0.8 x the initial loss, the final
        losses within 8 % of each other, the curves within 12 % of each other at every step; measured 4.6 % / 8.0 %).  A mode whose
        gradients were unusable fails here;
    (2) at the fp32-level run's final weights, ONE forward + backward in four arithmetics: the fp32-MFMA kernels as the control of
        the instrument (every conv-weight gradient at cosine >= 0.999 of the default arithmetic's; measured 0.99974 at worst -- two
        arithmetics that agree to 1e-6 per convolution already differ by 2 % in a gradient here), then ``bf16x1`` and ``bf16``:
        loss within 1 %, classifier gradient within 30 % relative L2 (measured 18 %), and the conv-weight gradients of bf16 storage
        no further from the fp32-level ones than 1.3 x what ``bf16x1`` on fp32 tensors already is;
    (3) a per-stage profile of the conv-weight gradients against the fp32-level ones, with floors a noise vector would miss by far
        (its cosine would be ~0): median cosine >= 0.6 in layer4 and >= 0.05 in every stage (measured 0.11-0.30 in conv1, depending on the summation order).
        The ``bf16x2`` arithmetic (two bf16 pieces per operand, three products; tests/test_bf16x2_gpu.py) rides along: median cosine
        >= 0.99 in every stage (measured 0.995 in the stem to 0.9993 in layer4, relative L2 0.04 - 0.10 -- three to four times the
        distance between the two fp32-level arithmetics).

    What the measurement says about the absolute bar the round-2 review asked for (cosine >= 0.9 / relative L2 <= 0.3 for EVERY
    conv weight): it CANNOT be met by bf16 operands on this network, conditioned problem or not -- also not by ``bf16x1``, whose
    tensors are fp32.  Measured here (median cosine per stage, bf16x1 / bf16 storage): layer4 0.75 / 0.72, layer3 0.40 / 0.33,
    layer2 0.34 / 0.27, layer1 0.32 / 0.25, stem 0.26 / 0.30; relative L2 0.75 - 1.2.  The control of part (2) says why: on this
    network a perturbation of 1e-6 per convolution (fp32-MFMA against the default arithmetic) is already amplified to 2 % of a
    gradient.  The gradient through ~50 train-mode BatchNorm + ReLU layers is discontinuous in the activations -- every
    pre-activation within the rounding error of zero takes the other ReLU branch, and each flip moves its channel's gradient by
    per cents (tests/test_model_gpu.py counts them for the fp32-level path: a few hundred of 1.5e8 signs) -- and each BatchNorm
    backward keeps only what is left after subtracting the batch mean and the xhat projection.  At 2^-9 relative rounding per bf16
    operand the flipped fraction is three orders of magnitude larger.  The rounding is unbiased, which is why the optimisation of
    part (1) still tracks the fp32-level run.  (Parity unpinned: the reference trains in precision 32, libs/cil/cil.py:744-756;
    the bars are this repository's.)"""
    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K
    from test_model_gpu import _report
    K_, B, STEPS = 11, 8, 40
    torch.manual_seed(5)
    cfg = O.r50_cfg(num_classes=K_, depth=50, head='SimpleLinear', loss='CrossEntropyLoss', dropout_ratio=0.0)
    gen = torch.Generator().manual_seed(17)
    imgs, labels = _structured_clips(B, K_, 224, gen)
    imgs, labels = imgs.to(dev), labels.to(dev)
    init = bd.build_model(copy.deepcopy(cfg)).to(dev)
    opt_cfg = dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0),
                   lr=0.01, momentum=0.9, weight_decay=1e-4)

    def train(mode):
        prev = K.set_conv_arith(mode)
        try:
            m = copy.deepcopy(init)
            K.bump_weight_epoch()
            m.train()
            engine = bd.TrainEngine(m, bd.build_optimizer(m, opt_cfg), grad_clip=1.0)
            curve = [engine.step(dict(imgs=imgs, label=labels))['loss_cls'].item() for _ in range(STEPS)]
            torch.cuda.synchronize()
            return m, curve
        finally:
            K.set_conv_arith('bf16x3')
            K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev
    base, curve = train('bf16x3')
    _, curve16 = train('bf16')
    assert curve[-1] < 0.8 * curve[0], curve                       # the fp32-level path has started to fit the batch
    assert curve16[-1] < 0.8 * curve16[0], curve16                 # and so has the bf16-storage path
    assert abs(curve16[-1] - curve[-1]) <= 8e-2 * curve[-1], (curve[-1], curve16[-1])
    worst_step = max(abs(a - b) / b for a, b in zip(curve16, curve))
    assert worst_step <= 12e-2, worst_step
    # the two-piece arithmetic, with the fp32-MFMA kernels as the yardstick: 40 steps of momentum SGD on eight clips amplify ANY
    # difference in arithmetic (the two fp32-level paths drift apart too), so the bar is the yardstick's own drift, not zero
    _, curve2 = train('bf16x2')
    _, curve_c = train('f32mfma')
    worst_step2 = max(abs(a - b) / b for a, b in zip(curve2, curve))
    worst_step_c = max(abs(a - b) / b for a, b in zip(curve_c, curve))
    _report(f'[bf16x2 training curve] fp32-level {curve[0]:.4f} -> {curve[-1]:.4f}, bf16x2 {curve2[0]:.4f} -> {curve2[-1]:.4f} (worst step apart '
            f'{worst_step2:.4f}), fp32-MFMA control -> {curve_c[-1]:.4f} (worst step apart {worst_step_c:.4f})')
    assert curve2[-1] < 0.8 * curve2[0] and worst_step2 <= max(12e-2, 3.0 * worst_step_c), (worst_step2, worst_step_c)

    losses, grads = {}, {}
    for mode in ('bf16x3', 'f32mfma', 'bf16x2', 'bf16x1', 'bf16'):
        prev = K.set_conv_arith(mode)
        try:
            m = copy.deepcopy(base)
            K.bump_weight_epoch()
            m.train()
            out = m(imgs, labels)
            out['loss_cls'].backward()
            torch.cuda.synchronize()
            losses[mode] = out['loss_cls'].item()
            grads[mode] = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            K.set_conv_arith('bf16x3')
            K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev

    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-300)).item()

    def cos(a, b):
        return ((a * b).sum() / (a.norm() * b.norm() + 1e-300)).item()
    control = min(cos(grads['f32mfma'][n], g) for n, g in grads['bf16x3'].items() if n.endswith('conv.weight') or n.endswith('net.weight'))
    assert control >= 0.999, control                           # the instrument itself: two fp32-level arithmetics agree (measured 0.99974)
    stages = ['conv1', 'layer1', 'layer2', 'layer3', 'layer4']
    prof = {}
    for mode in ('bf16x2', 'bf16x1', 'bf16'):
        for st in stages:
            names = [n for n in grads['bf16x3'] if n.startswith('backbone.' + st + '.') and (n.endswith('conv.weight') or n.endswith('net.weight'))]
            cs = sorted(cos(grads[mode][n], grads['bf16x3'][n]) for n in names)
            rs = sorted(rel(grads[mode][n], grads['bf16x3'][n]) for n in names)
            prof[(mode, st)] = (cs[len(cs) // 2], cs[0], rs[len(rs) // 2], rs[-1])
    _report(f'[config-5 gradient fidelity] loss curves fp32-level {curve[0]:.4f} -> {curve[-1]:.4f}, bf16 storage {curve16[0]:.4f} -> {curve16[-1]:.4f} '
            f'(worst step apart {worst_step:.3f}); control cosine fp32-MFMA vs default {control:.6f}; losses at the final weights {losses}; conv-weight gradients vs the fp32-level ones, per stage '
            f'(median cosine, min cosine, median relL2, max relL2): '
            + '; '.join(f'{mode} {st} ({a:.4f}, {b:.4f}, {c:.3f}, {d:.3f})' for (mode, st), (a, b, c, d) in prof.items()))
    assert all(torch.isfinite(g).all() for g in grads['bf16'].values())
    assert abs(losses['bf16'] - losses['bf16x3']) <= 1e-2 * abs(losses['bf16x3']), losses
    worst_ratio = 0.0
    for n, g in grads['bf16x3'].items():
        if n.startswith('cls_head'):
            assert rel(grads['bf16'][n], g) <= 0.3, (n, rel(grads['bf16'][n], g))
        elif n.endswith('conv.weight') or n.endswith('net.weight'):
            ratio = rel(grads['bf16'][n], g) / max(rel(grads['bf16x1'][n], g), 1e-3)
            worst_ratio = max(worst_ratio, ratio)
            assert ratio <= 1.3, (n, ratio)
    assert all(prof[('bf16x2', st)][0] >= 0.99 for st in stages), {st: prof[('bf16x2', st)] for st in stages}   # two pieces: measured 0.995 - 0.9993
    for mode in ('bf16x1', 'bf16'):
        assert prof[(mode, 'layer4')][0] >= 0.6, (mode, prof[(mode, 'layer4')])
        # every stage stays positively correlated; the early stages' value moves with the summation order of the fp32-level
        # trajectory that produced the weights (conv1: 0.30 with the shipped kernels, 0.11 with a 16x16x32-MFMA build of them)
        assert all(prof[(mode, st)][0] >= 0.05 for st in stages), {st: prof[(mode, st)] for st in stages}


def test_kd_step_on_bf16_features(dev, bf16_mode):
    """libs/cil/cil.py:512-556 with hooks on the four stages and the average pool: the hooked stage outputs are bf16 views, the
    KD-MSE kernels read them as such; every term within 10 % and the total within 5 % of the fp32 CPU oracle, gradients finite and fp32."""
    import bdvcil_amd as bd
    K_ = 11
    names = ['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool']
    weights, scale = [0.01] * 5, [1.0, 3.3466401061363023]
    torch.manual_seed(3)
    cfg = O.r50_cfg(num_classes=K_, depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
    ref, mod = O.build_model(copy.deepcopy(cfg)), bd.build_model(copy.deepcopy(cfg))
    mod.load_state_dict(ref.state_dict())
    torch.manual_seed(5)
    ref_prev, mod_prev = O.build_model(copy.deepcopy(cfg)), bd.build_model(copy.deepcopy(cfg))
    mod_prev.load_state_dict(ref_prev.state_dict())
    mod.to(dev); mod_prev.to(dev)
    gen = torch.Generator().manual_seed(11)
    imgs, labels = torch.randn(4, 8, 3, 64, 64, generator=gen), torch.randint(0, K_, (4, 1), generator=gen)
    rt, rpt = O.FeatureTap(ref, names), O.FeatureTap(ref_prev, names)
    ref.train(); ref_prev.eval()
    rl = O.kd_training_step(ref, ref_prev, rt, rpt, imgs, labels, names, weights, scale[1], True)
    ch, ph = bd.OutputHook(mod, names), bd.OutputHook(mod_prev, names)
    mod.train(); mod_prev.eval()
    ol = bd.base_training_step(mod, dict(imgs=imgs.to(dev), label=labels.to(dev)), current_task=1, prev_model=mod_prev,
                               current_hooks=ch, prev_hooks=ph, kd_modules_names=names, kd_weight_by_module=weights,
                               adaptive_scale_factors=scale)
    ol['loss'].backward()
    assert ch.get_layer_output('backbone.layer2').dtype == BF and ch.get_layer_output('cls_head.avg_pool').dtype == torch.float32
    for n in names:
        assert abs(ol[n].item() - rl[n].item()) <= 1e-1 * max(1e-3, abs(rl[n].item())), (n, ol[n].item(), rl[n].item())
    assert abs(ol['loss'].item() - rl['loss'].item()) <= 5e-2 * max(1.0, abs(rl['loss'].item()))
    for p in mod.parameters():
        assert p.grad is None or (p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()))


def test_task_loop_runs_in_bf16_storage(tmp_path, dev, bf16_mode):
    """Two tasks of the CIL loop (feature KD on hooked bf16 stage outputs, exemplar selection and class means from the fp32
    representations, the files of libs/cil/cil.py:28-375) with bf16-stored activations: finite losses, every file of a run."""
    import os

    import numpy as np
    from test_task_loop_gpu import BUDGET, TASKS, _config, _loop
    cfg = _config(tmp_path, ending_task=1, num_epochs_per_task=1)
    loop = _loop(cfg)
    hist = loop.train()
    assert len(hist) == 2 and all(np.isfinite(h['train_loss']).all() for h in hist)
    work = cfg['work_dir']
    for t in range(2):
        assert os.path.exists(os.path.join(work, 'ckpt', f'ckpt_task_{t}.pt'))
        assert os.path.exists(os.path.join(work, 'exemplar', f'exemplar_task_{t}.txt'))
    lines = open(os.path.join(work, 'exemplar', 'exemplar_task_1.txt')).read().split('\n')
    assert len([l for l in lines if l.strip()]) == BUDGET * len(TASKS[1])          # the new classes of task 1
    sd = torch.load(os.path.join(work, 'ckpt', 'ckpt_task_1.pt'), weights_only=True)
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())       # checkpoints are the same fp32 state_dicts
