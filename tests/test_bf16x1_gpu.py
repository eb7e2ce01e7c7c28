"""The reduced-precision conv arithmetic of BASELINE config 5 ("TSM-ResNet50 fp16 with MFMA fp16 tiles ... batch 64"):
``set_conv_arith('bf16x1')`` = every conv operand rounded to bf16 (the hi plane of the bf16-piece kernels), ONE
v_mfma_f32_32x32x16_bf16 product per step, fp32 accumulate, fp32 tensors in HBM.

The reference trains in precision 32 (libs/cil/cil.py:744-756): there are no reference numerics for this mode.  Stated bars:
  * kernels: the result equals, to fp32 summation order (2e-5 of scale), the fp32 convolution of the bf16-ROUNDED operands --
    the definition of the mode -- and lies within 1e-2 of the scale of the full-precision result;
  * model: eval logits within 5e-2 of the fp32 CPU oracle on the small test nets, arg-max allowed to differ only where the
    oracle's top-2 margin is below 5e-2; a short training run follows the oracle's loss curve within 5 % and goes down."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tsm_oracle as O
from oracle.tsm_oracle import temporal_shift

pytestmark = pytest.mark.gpu

CASES = [
    (16, 14, 14, 128, 256, 1, 1, 0, 8, 16),
    (8, 9, 9, 128, 256, 3, 1, 1, 1, 0),
    (8, 8, 8, 256, 512, 3, 2, 1, 8, 32),
    (16, 7, 7, 256, 128, 1, 1, 0, 8, 32),
    (8, 12, 12, 64, 64, 3, 1, 1, 1, 0),
    (8, 8, 8, 256, 256, 1, 2, 0, 1, 0),
]


@pytest.fixture
def bf16x1():
    from bdvcil_amd import kernels as K
    prev = K.set_conv_arith('bf16x1')
    yield
    K.set_conv_arith('bf16x3')
    K.FPROP_X3, K.DGRAD_X3, K.WGRAD_X3 = prev


def _r(t):
    return t.to(torch.bfloat16).float()


def _err(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize('case', CASES)
def test_kernels_equal_conv_of_rounded_operands(case, dev, bf16x1):
    from bdvcil_amd import kernels as K
    N, H, W, Cin, Cout, R, st, pad, T, fold = case
    gen = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, R, R, generator=gen) / (Cin * R * R) ** 0.5
    dy = None

    def run(xx, ww, dd):
        xx = xx.clone().requires_grad_(True)
        ww = ww.clone().requires_grad_(True)
        xs = temporal_shift(xx, T, Cin // fold) if fold > 0 else xx
        y = F.conv2d(xs, ww, stride=st, padding=pad)
        return xx, ww, y
    xf, wf, yf = run(x, w, None)
    dy = torch.randn(yf.shape, generator=gen)
    yf.backward(dy)
    # what the mode computes: fprop on rounded x, w; dgrad on rounded dy, w; wgrad on rounded dy, x
    _, _, y_r = run(_r(x), _r(w), None)
    xr, wr, y_tmp = run(x, _r(w), None)
    y_tmp.backward(_r(dy))
    dx_r = xr.grad
    xr2, wr2, y_tmp2 = run(_r(x), w, None)
    y_tmp2.backward(_r(dy))
    dw_r = wr2.grad
    g = K.make_geom(N, H, W, Cin, Cout, R, R, st, pad, T, fold)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    y = K.conv_fprop(xd, wd, g).cpu().permute(0, 3, 1, 2)
    dx = K.conv_dgrad(dyd, wd, g).cpu().permute(0, 3, 1, 2)
    dw = K.conv_wgrad(dyd, xd, g).cpu().permute(0, 3, 1, 2)
    assert _err(y, y_r.detach()) <= 2e-5 and _err(dx, dx_r) <= 2e-5 and _err(dw, dw_r) <= 2e-5
    # and how far that is from full precision
    for a, b in ((y, yf.detach()), (dx, xf.grad), (dw, wf.grad)):
        assert _err(a, b) <= 1e-2
    assert 1e-4 <= _err(y, yf.detach())          # it really is the reduced arithmetic


def test_model_logits_and_short_training_run(dev, bf16x1):
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
    assert (o - r).abs().max().item() <= 5e-2, (o - r).abs().max().item()
    top2 = r.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 5e-2
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
        assert abs(a - b) <= 5e-2 * max(1.0, abs(b)), (hip_curve, ref_curve)
