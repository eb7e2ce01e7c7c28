"""``set_conv_arith('bf16x2')``: every conv operand cut to its TWO leading bf16 pieces (hi + mid = 16 significand bits) and three
v_mfma_f32_32x32x16_bf16 products per step (hi*hi + hi*mid + mid*hi), fp32 accumulate, fp32 tensors in HBM -- half the matrix work
of the default six-product arithmetic.  An option, never the default and never bench.py's `value` (it reports it under
`alt_arith`).  The reference computes in fp32 on the CPU and, on its GPUs, in TF32 unless told otherwise (torch's cudnn.allow_tf32
default): this arithmetic sits between the two, so there are no reference numerics for it.  Stated bars:
  * kernels (fprop / dgrad / wgrad through the plane kernels; sites they do not cover keep three pieces): the result equals, to
    fp32 accumulation (2e-6 of scale), the exact value of its definition conv(hi + mid, hi + mid) - conv(mid, mid) computed in
    fp64 on the CPU; it lies within 2e-5 of the scale of the fp64 convolution of the unsplit operands (measured 4 - 6e-6) and
    further than 2e-6 from it (i.e. the mode really ran: the default arithmetic is at 2 - 6e-7);
  * model: eval logits within the north-star bar of 1e-3 of the fp32 CPU oracle AND within 1e-5 (measured 2 - 4e-7), arg-max equal;
    a short training run follows the oracle's loss curve within 1e-3;
  * gradients of a conditioned R50 problem: tests/test_bf16_storage_gpu.py::test_r50_gradient_fidelity_on_a_conditioned_problem
    carries this mode beside the others (median cosine to the fp32-level gradient >= 0.99 in every stage)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import tsm_oracle as O
from oracle.tsm_oracle import temporal_shift

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, R, stride, pad, T, fold)
CASES = [
    (16, 14, 14, 128, 256, 1, 1, 0, 8, 16),     # a block's conv1 with the temporal shift: 128x128 four-wave tile in this mode
    (8, 9, 9, 128, 256, 3, 1, 1, 1, 0),
    (8, 8, 8, 256, 512, 3, 2, 1, 8, 32),
    (16, 7, 7, 256, 128, 1, 1, 0, 8, 32),
    (8, 12, 12, 64, 64, 3, 1, 1, 1, 0),
    (8, 8, 8, 256, 256, 1, 2, 0, 1, 0),
    (32, 14, 14, 1024, 256, 1, 1, 0, 1, 0),     # long K, K-split remainder tiles
    (32, 14, 14, 256, 1024, 1, 1, 0, 1, 0),
    (8, 28, 28, 512, 128, 1, 1, 0, 8, 64),
]


@pytest.fixture
def bf16x2():
    from bdvcil_amd import kernels as K
    prev = K.set_conv_arith('bf16x2')
    yield
    K.set_conv_arith('bf16x3')
    K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


def _two(t):
    hi = t.to(torch.bfloat16).float()
    mid = (t - hi).to(torch.bfloat16).float()
    return hi.double(), mid.double()


def _err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize('case', CASES)
def test_kernels_equal_their_definition(case, dev, bf16x2):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    gen = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, R, R, generator=gen) / (Cin * R * R) ** 0.5
    sh = (lambda t: temporal_shift(t, T, Cin // fold)) if fold > 0 else (lambda t: t)

    def triple(xx, ww, dd):
        """(y, dx, dw) of the shifted convolution for the given x, w and (for the gradients) dy, in fp64."""
        xx = xx.clone().requires_grad_(True)
        ww = ww.clone().requires_grad_(True)
        y = F.conv2d(sh(xx), ww, stride=st, padding=pad)
        if dd is None:
            return y.detach(), None, None
        y.backward(dd)
        return y.detach(), xx.grad, ww.grad
    y64, _, _ = triple(x.double(), w.double(), None)
    dy = torch.randn(y64.shape, generator=gen)
    _, dx64, dw64 = triple(x.double(), w.double(), dy.double())
    (xh, xm), (wh, wm), (dh, dm) = _two(x), _two(w), _two(dy)
    # the definition, operand pair by operand pair: fprop (x, w), dgrad (dy, w), wgrad (dy, x)
    y_def = triple(xh + xm, wh + wm, None)[0] - triple(xm, wm, None)[0]
    dx_def = triple(x.double(), wh + wm, dh + dm)[1] - triple(x.double(), wm, dm)[1]
    dw_def = triple(xh + xm, w.double(), dh + dm)[2] - triple(xm, w.double(), dm)[2]
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    assert K.conv_kernel_name(g, 'fprop').replace(' ', '').endswith(',2>'), K.conv_kernel_name(g, 'fprop')
    assert K.conv_kernel_name(g, 'dgrad').replace(' ', '').endswith(',2>'), K.conv_kernel_name(g, 'dgrad')
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    y = K.conv_fprop(xd, wd, g).cpu().permute(0, 3, 1, 2).double()
    dx = K.conv_dgrad(dyd, wd, g).cpu().permute(0, 3, 1, 2).double()
    dw = K.conv_wgrad(dyd, xd, g).cpu().permute(0, 3, 1, 2).double()
    for name, got, want, full in (('y', y, y_def, y64), ('dx', dx, dx_def, dx64), ('dw', dw, dw_def, dw64)):
        assert _err(got, want) <= 2e-6, (name, _err(got, want))
        assert 2e-6 <= _err(got, full) <= 2e-5, (name, _err(got, full))


@pytest.mark.parametrize('depth,S', [(18, 64), (50, 224)])
def test_eval_logits(depth, S, dev, bf16x2):
    import bdvcil_amd as bd
    torch.manual_seed(3)
    cfg = O.r50_cfg(num_classes=11, depth=depth, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.0)
    ref = O.build_model(copy.deepcopy(cfg))
    mod = bd.build_model(copy.deepcopy(cfg))
    mod.load_state_dict(ref.state_dict())
    mod.to(dev)
    imgs = torch.randn(2, 8, 3, S, S, generator=torch.Generator().manual_seed(11))
    ref.eval(); mod.eval()
    with torch.no_grad():
        for mode in ('prob', 'score'):
            ref.test_cfg['average_clips'] = mod.test_cfg['average_clips'] = mode
            r = ref.forward_test(imgs)
            o = mod.forward_test(imgs.to(dev)).cpu()
            e = (o - r).abs().max().item()
            assert e <= 1e-3 and e <= 1e-5, (mode, e)             # the north star's bar, and this arithmetic's own
            assert torch.equal(o.argmax(1), r.argmax(1))


def test_short_training_run_follows_the_oracle(dev, bf16x2):
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
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (hip_curve, ref_curve)


def test_i3d_eval_logits(dev, bf16x2):
    """The arithmetic reaches the I3D blocks too (k x 1 temporal sites and per-frame sites run the same plane kernels): I3D-R50 eval
    logits at 8 x 64 x 64, arg-max equal.  The I3D head is a plain linear layer on features of a random-init network evaluated with
    fresh running statistics, so the logits are not O(1) (largest: 90.7): the bar is relative, 1e-4 of the largest logit (measured 1.4e-3 absolute = 1.6e-5 relative; probabilities: 1e-5)."""
    from test_i3d_gpu import _pair
    ref, mod = _pair(9, dev)
    imgs = torch.randn(2, 2, 3, 8, 64, 64, generator=torch.Generator().manual_seed(3))
    ref.eval(); mod.eval()
    with torch.no_grad():
        for mode in ('prob', 'score'):
            ref.test_cfg['average_clips'] = mod.test_cfg['average_clips'] = mode
            r = ref(imgs, return_loss=False)
            o = mod(imgs.to(dev), return_loss=False).cpu()
            e, scale = (o - r).abs().max().item(), max(1.0, r.abs().max().item())
            assert e <= (1e-5 if mode == 'prob' else 1e-4 * scale), (mode, e, scale)
            assert torch.equal(o.argmax(1), r.argmax(1))
