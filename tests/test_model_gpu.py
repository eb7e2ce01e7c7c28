"""End-to-end parity of the HIP recognizer against the CPU oracle on identical seeded clips and weights.

Bars (BASELINE.json north_star): logits within 1e-3 (fp32), label indices bit-exact; gradients / updated
parameters within 1e-3 of their scale.  Dropout is 0 in parity runs (RNG streams differ by design)."""
import copy

import numpy as np

import pytest
import torch
import torch.nn.functional as F

from oracle import tsm_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_only(depth, head, loss, K=11, seed=0, nb_proxies=1, want_cfg=False):
    torch.manual_seed(seed)
    cfg = O.r50_cfg(num_classes=K, depth=depth, head=head, loss=loss, dropout_ratio=0.0, nb_proxies=nb_proxies)
    ref = O.build_model(copy.deepcopy(cfg))
    # perturb BN affine/running stats so that they matter
    g = torch.Generator().manual_seed(seed + 1)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
    return (ref, cfg) if want_cfg else ref


def _pair(depth, head, loss, K=11, dev=None, seed=0, nb_proxies=1):
    import bdvcil_amd as bd
    ref, cfg = _oracle_only(depth, head, loss, K, seed, nb_proxies, want_cfg=True)
    mod = bd.build_model(copy.deepcopy(cfg))
    mod.load_state_dict(ref.state_dict())
    return ref, mod.to(dev), cfg


class ReluRecorder:
    """Records the argument of every F.relu call of the oracle's backbone, in execution order (stem, then per block:
    conv1, [conv2,] block output).  The oracle calls ``F.relu`` through its module-level ``F`` (oracle/tsm_oracle.py)."""

    class _Proxy:
        def __init__(self, real, sink):
            self._real, self._sink = real, sink

        def relu(self, x, *a, **kw):
            if x.dim() == 4:
                self._sink.append(x.detach())
            return self._real.relu(x, *a, **kw)

        def __getattr__(self, name):
            return getattr(self._real, name)

    def __enter__(self):
        self.pre, self._saved = [], O.F
        O.F = ReluRecorder._Proxy(self._saved, self.pre)
        return self

    def __exit__(self, *exc):
        O.F = self._saved
        return False


def relu_site_owners(ref):
    """Parameter-name prefixes of the conv+BN units whose output feeds each recorded ReLU, in the recorder's order."""
    sites = [['backbone.conv1.']]
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(ref.backbone, f'layer{li}')):
            pre = f'backbone.layer{li}.{bi}.'
            names = [n for n in ('conv1', 'conv2', 'conv3') if hasattr(blk, n)]
            for n in names[:-1]:
                sites.append([pre + n + '.'])
            sites.append([pre + names[-1] + '.', pre + 'downsample.'])
    return sites


def _report(line):
    """Parity figures of passing tests are kept (pytest -q hides prints): appended to gpurun_out/parity_report.log."""
    import os
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_report.log'), 'a') as f:
            f.write(line.strip() + '\n')


def count_relu_flips(pre64, pre32, taps):
    """Per ReLU site: number of elements whose sign bit in the HIP mask differs from the fp64 oracle's ``pre > 0``.
    A difference is legitimate only where fp32 rounding can reach zero: the fp64 pre-activation must lie within
    max(1e-5, 3 x the largest deviation of the fp32 CPU oracle's own pre-activations at that site) of zero, both measured
    in units of the channel's largest magnitude (the CPU oracle's deviation grows with depth: ~2e-6 at the stem, ~2e-5
    behind 40 conv+BN layers).  Returns (flips per site, flips of the fp32 CPU oracle per site)."""
    assert len(pre64) == len(taps) == len(pre32), (len(pre64), len(pre32), len(taps))
    flips, flips32 = [], []
    for k, (pre, p32, (shape, mask)) in enumerate(zip(pre64, pre32, taps)):
        N, H, W, C = shape
        assert tuple(pre.shape) == (N, C, H, W), (k, tuple(pre.shape), shape)
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder='little').astype(bool).reshape(N, H, W, C)
        want = (pre > 0).permute(0, 2, 3, 1).numpy()
        diff = torch.from_numpy(bits != want)
        n = int(diff.sum())
        cmax = pre.abs().amax(dim=(0, 2, 3), keepdim=True)
        noise32 = float(((p32.double() - pre).abs() / cmax).max())
        flips32.append(int(((p32 > 0) != (pre > 0)).sum()))
        if n:
            rel = (pre.abs() / cmax).permute(0, 2, 3, 1)[diff]
            bar = max(1e-5, 3 * noise32)
            assert float(rel.max()) <= bar, (f'ReLU site {k}: a sign differs where the fp64 pre-activation is {float(rel.max()):.2e} of '
                                              f'its channel max (fp32 CPU oracle deviates by up to {noise32:.2e} there)')
        flips.append(n)
    return flips, flips32


def count_pool_flips(pre64, pre32, idx):
    """The stem max-pool's arg-max is as discontinuous as a ReLU sign: two candidates of a 3x3 window within fp32 rounding of
    each other can be picked either way, and the stem's gradients then differ by that element's share.  Number of output
    elements whose arg-max (the HIP path's 3 r + s code) differs from the fp64 oracle's, windows whose maximum is <= 0 aside
    (ReLU masks their gradient wherever it lands).  A difference is legitimate only where the two candidates are within
    max(1e-5, 3 x the fp32 CPU oracle's own deviation) of the channel's scale in the fp64 oracle."""
    import torch.nn.functional as F
    act = pre64.clamp_min(0)
    N, C, H, W = act.shape
    m, flat = F.max_pool2d(act, 3, 2, 1, return_indices=True)
    Ho, Wo = m.shape[2:]
    h, w = flat // W, flat % W
    ho = torch.arange(Ho).view(1, 1, Ho, 1)
    wo = torch.arange(Wo).view(1, 1, 1, Wo)
    code = (h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1))
    got = idx.cpu().view(N, Ho, Wo, C).permute(0, 3, 1, 2).long()
    diff = (code != got) & (m > 0)
    n = int(diff.sum())
    if n:
        cmax = pre64.abs().amax(dim=(0, 2, 3), keepdim=True)
        noise32 = float(((pre32.double() - pre64).abs() / cmax).max())
        r, s_ = got // 3, got % 3
        hh = (2 * ho - 1 + r).clamp(0, H - 1)
        ww = (2 * wo - 1 + s_).clamp(0, W - 1)
        picked = act.flatten(2).gather(2, (hh * W + ww).flatten(2)).view_as(m)
        gap = ((m - picked) / cmax)[diff]
        bar = max(1e-5, 3 * noise32)
        assert float(gap.max()) <= bar, f'stem max-pool: an arg-max differs where the candidates are {float(gap.max()):.2e} of the channel scale apart'
    return n


FLIP_FREE_SEED = 72     # chosen with tools/find_flip_free_seed.py (ReLU signs and stem max-pool arg-max; see test_train_step)


def _clips(B, T, S, K, seed=0):
    g = torch.Generator().manual_seed(100 + seed)
    return torch.randn(B, T, 3, S, S, generator=g), torch.randint(0, K, (B, 1), generator=g)


def _rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def _rel_l2(a, b):
    """Relative L2 error.  Used for gradients: train-mode BN + ReLU masks (and max-pool arg-max) are
    discontinuous, so a handful of activations within ~1e-6 of zero take different branches in any two fp32
    implementations (the fp32 CPU oracle differs from its own fp64 run in the same way);
    each flip perturbs one channel's gradient by a few per cent of its scale.  A max-abs bar would measure
    those rare flips, not the kernels; the kernels themselves are held to 2e-5 in test_conv_gpu / test_ops_gpu."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize('depth', [18, 50])
def test_train_step_is_bit_identical_with_and_without_the_apply_passes(depth, dev, monkeypatch):
    """BDVCIL_PRE_BN=1 (off by default: measured 0.6 ms per step slower; it saves the memory of the
    activations between conv1 / conv2 / conv3): the BatchNorm + ReLU of conv1 / conv2 applied in the loaders of the conv that
    consumes them, their activation and mask never written, ReLU signs derived from the conv output in the backward kernels.
    Both weight-gradient forms (activation recomputed / BatchNorm in the loader).  R18: not a bit of the training step changes."""
    from bdvcil_amd import functional as Fn
    from bdvcil_amd import kernels as KK
    monkeypatch.setattr(KK, 'PRE_BN_1X1_ONLY', False)       # 3x3 consumers too (R18 has no others)
    ref, mod, _ = _pair(depth, 'LocalSimilarityClassifier', 'LSCLoss', K=11, dev=dev)
    state = copy.deepcopy(mod.state_dict())
    imgs, labels = _clips(2, 8, 64, 11, seed=3)
    results = []
    prev = Fn.PRE_BN
    try:
        for flag, wg in ((True, 'recompute'), (False, 'recompute'), (True, 'loader')):
            Fn.PRE_BN, Fn.PRE_BN_WGRAD = flag, wg
            mod.load_state_dict(state)
            mod.zero_grad(set_to_none=True)
            mod.train()
            out = mod(imgs.to(dev), labels.to(dev), batch_data=None)
            out['loss_cls'].backward()
            torch.cuda.synchronize()
            results.append((out['loss_cls'].detach().clone(), {n: p.grad.detach().clone() for n, p in mod.named_parameters() if p.grad is not None},
                            {n: b.detach().clone() for n, b in mod.named_buffers()}))
    finally:
        Fn.PRE_BN, Fn.PRE_BN_WGRAD = prev, 'recompute'
    assert torch.equal(results[0][0], results[2][0]) and all(torch.equal(results[0][1][n], results[2][1][n]) for n in results[0][1])
    (l1, g1, b1), (l0, g0, b0) = results[0], results[1]
    assert g1.keys() == g0.keys()
    if depth == 18:         # every consumer keeps the kernel the planner picks anyway: not a bit changes
        assert torch.equal(l1, l0)
        for n in g1:
            assert torch.equal(g1[n], g0[n]), n
        for n in b1:
            assert torch.equal(b1[n], b0[n]), n
    else:                   # R50: some 1x1 consumers move to another tile (another summation order): fp32 rounding, which a
        #                     ReLU / max-pool decision within 1e-6 of a tie turns into a per-cent change of that channel's share
        assert abs(l1.item() - l0.item()) <= 1e-6 * abs(l0.item())
        for n in g1:
            bar = 1e-4 if n.startswith(('cls_head.', 'backbone.layer4.2.')) else 1e-2
            assert _rel_l2(g1[n], g0[n]) <= bar, (n, _rel_l2(g1[n], g0[n]))
        for n in b1:
            if not n.endswith('num_batches_tracked'):
                assert _rel(b1[n], b0[n]) <= 1e-5, n


@pytest.mark.parametrize('depth,S', [(18, 64), (34, 64), (50, 64), (50, 224)])
def test_eval_logits(depth, S, dev, conv_arith):
    ref, mod, _ = _pair(depth, 'LocalSimilarityClassifier', 'LSCLoss', dev=dev)
    imgs, _ = _clips(2, 8, S, 11)
    ref.eval(); mod.eval()
    with torch.no_grad():
        for mode in ('prob', 'score'):
            ref.test_cfg['average_clips'] = mode
            mod.test_cfg['average_clips'] = mode
            r = ref.forward_test(imgs)
            o = mod.forward_test(imgs.to(dev)).cpu()
            assert (o - r).abs().max().item() <= 1e-3, (mode, (o - r).abs().max().item())
            assert torch.equal(o.argmax(1), r.argmax(1))


@pytest.mark.parametrize('depth,head,loss,S,B,clip_seed', [
    (18, 'LocalSimilarityClassifier', 'LSCLoss', 224, 4, 0),      # BASELINE config 1 (R18, B=4)
    (18, 'LocalSimilarityClassifier', 'LSCLoss', 64, 2, FLIP_FREE_SEED),   # no activation of the fp64 oracle within 1e-6 std of 0
    (34, 'LocalSimilarityClassifier', 'LSCLoss', 64, 2, 0),
    (50, 'SimpleLinear', 'CrossEntropyLoss', 64, 2, 0),
    (50, 'LocalSimilarityClassifier', 'LSCLoss', 224, 2, 0),
])
def test_train_step(depth, head, loss, S, B, clip_seed, dev, conv_arith):
    """Loss / accuracies / every parameter gradient / BN statistics / one clipped SGD step.

    Gradients are judged against the fp64 run of the oracle.  Train-mode BN + ReLU is discontinuous: an activation that the
    fp64 oracle places within ~1e-6 of zero can take the other branch in any fp32 implementation (the fp32 CPU oracle does
    it too), and one flipped element moves its channel's gradient by per cents.  So the ReLU sign bits the HIP path wrote
    are read back and compared with the oracle's: every difference must sit where the fp64 pre-activation is within 1e-5
    of its channel's scale (largest magnitude) of zero, and
      * parameters that no flipped ReLU lies behind (in backward order) are held to  relL2 <= 3 * e_f32 + 1e-4,
      * parameters upstream of a counted flip keep the looser  3 * e_f32 + 1e-2."""
    import bdvcil_amd as bd
    from bdvcil_amd import functional as Fn
    K = 11
    ref, mod, _ = _pair(depth, head, loss, K=K, dev=dev)
    ref64 = copy.deepcopy(ref).double()
    imgs, labels = _clips(B, 8, S, K, seed=clip_seed)
    ref.train(); mod.train(); ref64.train()
    with ReluRecorder() as rec32:
        rl = ref(imgs, labels)
    rl['loss_cls'].backward()
    with ReluRecorder() as rec:
        r64 = ref64(imgs.double(), labels)
    r64['loss_cls'].backward()
    Fn.RELU_MASK_TAP = taps = []
    Fn.POOL_IDX_TAP = pool_idx = []
    try:
        ol = mod(imgs.to(dev), labels.to(dev), batch_data=None)
    finally:
        Fn.RELU_MASK_TAP = Fn.POOL_IDX_TAP = None
    ol['loss_cls'].backward()
    assert abs(ol['loss_cls'].item() - rl['loss_cls'].item()) <= 1e-4 * max(1.0, abs(rl['loss_cls'].item()))
    assert abs(ol['top1_acc'].item() - rl['top1_acc'].item()) < 1e-6
    assert abs(ol['top5_acc'].item() - rl['top5_acc'].item()) < 1e-6
    flips, flips32 = count_relu_flips(rec.pre, rec32.pre, taps)
    owners = relu_site_owners(ref)
    assert len(owners) == len(flips)
    last_flip = max([k for k, n in enumerate(flips) if n], default=-1)
    pool_flips = count_pool_flips(rec.pre[0], rec32.pre[0], pool_idx[0])
    if pool_flips:
        last_flip = max(last_flip, 0)          # the stem unit (site 0) lies behind its max-pool
    _report(f'[relu flips] R{depth} S={S} B={B} {conv_arith}: {sum(flips)} of {sum(p.numel() for p in rec.pre)} signs differ '
            f'from the fp64 oracle at sites {[k for k, n in enumerate(flips) if n]} (fp32 CPU oracle: {sum(flips32)} at '
            f'{[k for k, n in enumerate(flips32) if n]}); stem max-pool arg-max differs at {pool_flips} elements')

    def behind_a_flip(name):           # a flip at site k perturbs the gradients of every unit up to and including site k
        for k, prefixes in enumerate(owners):
            if any(name.startswith(q) for q in prefixes):
                return k <= last_flip
        return False                   # head / loss parameters: after every ReLU of the backbone
    rp, r64p, op = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(mod.named_parameters())
    worst = {}
    for name, p in rp.items():
        if p.grad is None:
            assert op[name].grad is None or op[name].grad.abs().max().item() == 0, name
            continue
        assert op[name].grad is not None, name
        e_hip = _rel_l2(op[name].grad, r64p[name].grad)
        e_f32 = _rel_l2(p.grad, r64p[name].grad)
        loose = behind_a_flip(name)
        floor = 1e-2 if loose else 1e-4
        assert e_hip <= 3 * e_f32 + floor, (name, e_hip, e_f32, 'behind a flipped ReLU' if loose else 'no flip behind it', flips)
        if e_hip > worst.get(loose, (0,))[0]:
            worst[loose] = (e_hip, e_f32, name)
    _report(f'[grad parity] R{depth} S={S} B={B} {conv_arith}: worst (relL2 hip, relL2 fp32 CPU, name) with no flip behind: '
            f'{worst.get(False)}; behind a flip: {worst.get(True)}')
    # BN running statistics
    rb, ob = dict(ref.named_buffers()), dict(mod.named_buffers())
    for name, b in rb.items():
        if name.endswith('num_batches_tracked'):
            assert int(ob[name].item()) == int(b.item()), name
        else:
            assert _rel(ob[name], b) <= 1e-4, name
    # one optimizer step with clipping: updated parameters agree
    opt_cfg = dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0),
                   lr=0.01, momentum=0.9, weight_decay=1e-4)
    oopt = bd.build_optimizer(mod, opt_cfg)
    ropt = O.build_sgd(ref)
    before = {n: p.detach().clone() for n, p in rp.items()}
    torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 1.0)
    ropt.step()
    oopt.clip_grad_norm_(1.0)
    oopt.step()
    for name, p in rp.items():
        assert _rel(op[name], p) <= 1e-4 or _rel_l2(op[name].detach().cpu() - before[name], p.detach() - before[name]) <= 5e-2, name


def test_kd_step_and_hooks(dev):
    """libs/cil/cil.py:512-556 with the config's kd_modules_names / weights."""
    import bdvcil_amd as bd
    K = 11
    names = ['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool']
    weights = [0.01] * 5
    scale = [1.0, 3.3466401061363023]
    ref, mod, cfg = _pair(50, 'LocalSimilarityClassifier', 'LSCLoss', K=K, dev=dev)
    ref_prev, mod_prev, _ = _pair(50, 'LocalSimilarityClassifier', 'LSCLoss', K=K, dev=dev, seed=5)
    imgs, labels = _clips(2, 8, 64, K)
    rt, rpt = O.FeatureTap(ref, names), O.FeatureTap(ref_prev, names)
    ref.train()
    rl = O.kd_training_step(ref, ref_prev, rt, rpt, imgs, labels, names, weights, scale[1], True)
    rl['loss'].backward()
    ch, ph = bd.OutputHook(mod, names), bd.OutputHook(mod_prev, names)
    mod.train()
    ol = bd.base_training_step(mod, dict(imgs=imgs.to(dev), label=labels.to(dev)), current_task=1, prev_model=mod_prev,
                               current_hooks=ch, prev_hooks=ph, kd_modules_names=names, kd_weight_by_module=weights,
                               adaptive_scale_factors=scale)
    ol['loss'].backward()
    assert tuple(ch.get_layer_output('backbone.layer1').shape) == tuple(rt.out['backbone.layer1'].shape)
    assert tuple(ch.get_layer_output('cls_head.avg_pool').shape) == tuple(rt.out['cls_head.avg_pool'].shape)
    for n in names:
        assert abs(ol[n].item() - rl[n].item()) <= 1e-4 * max(1e-3, abs(rl[n].item())), n
    assert abs(ol['loss'].item() - rl['loss'].item()) <= 1e-4 * max(1.0, abs(rl['loss'].item()))
    rp, op = dict(ref.named_parameters()), dict(mod.named_parameters())
    for name in ['backbone.conv1.conv.weight', 'backbone.layer1.0.conv1.conv.net.weight', 'backbone.layer2.0.downsample.conv.weight',
                 'backbone.layer4.2.conv3.conv.weight', 'backbone.layer3.1.conv2.bn.weight', 'cls_head.fc_cls.weights',
                 'cls_head.loss_cls.eta']:
        assert _rel_l2(op[name].grad, rp[name].grad) <= 2e-2, name


@pytest.mark.parametrize('old_rows', [[0], [0, 1], []])
def test_kd_step_exemplar_only(old_rows, dev):
    """kd_exemplar_only=True (cil.py:529-536): the KD terms use only the tensor rows at the batch positions of old-class
    samples (one, several, none)."""
    import bdvcil_amd as bd
    K, prevK = 11, 5
    names = ['backbone.layer2', 'backbone.layer4', 'cls_head.avg_pool']
    ref, mod, cfg = _pair(18, 'LocalSimilarityClassifier', 'LSCLoss', K=K, dev=dev)
    ref_prev, mod_prev, _ = _pair(18, 'LocalSimilarityClassifier', 'LSCLoss', K=K, dev=dev, seed=5)
    imgs, labels = _clips(2, 8, 64, K)
    labels[:, 0] = torch.tensor([7, 9])
    for r in old_rows:
        labels[r, 0] = r + 1
    rt, rpt = O.FeatureTap(ref, names), O.FeatureTap(ref_prev, names)
    ref.train()
    rl = O.kd_training_step(ref, ref_prev, rt, rpt, imgs, labels, names, [0.5] * 3, 2.0, True, True, prevK)
    rl['loss'].backward()
    ch, ph = bd.OutputHook(mod, names), bd.OutputHook(mod_prev, names)
    mod.train()
    ol = bd.base_training_step(mod, dict(imgs=imgs.to(dev), label=labels.to(dev)), current_task=1, prev_model=mod_prev,
                               current_hooks=ch, prev_hooks=ph, kd_modules_names=names, kd_weight_by_module=[0.5] * 3,
                               adaptive_scale_factors=[1.0, 2.0], kd_exemplar_only=True, previous_task_num_classes=prevK)
    ol['loss'].backward()
    for n in names:
        a, b = (float(v.detach()) if torch.is_tensor(v) else float(v) for v in (ol[n], rl[n]))
        assert abs(a - b) <= 1e-4 * max(1e-3, abs(b)), n
        assert (b == 0) == (not old_rows)
    assert abs(ol['loss'].item() - rl['loss'].item()) <= 1e-4 * max(1.0, abs(rl['loss'].item()))
    rp, op = dict(ref.named_parameters()), dict(mod.named_parameters())
    for name in ['backbone.conv1.conv.weight', 'backbone.layer4.1.conv2.conv.weight', 'cls_head.fc_cls.weights']:
        assert _rel_l2(op[name].grad, rp[name].grad) <= 2e-2, name


def test_icarl_step(dev):
    import bdvcil_amd as bd
    K, prevK = 11, 6
    ref, mod, _ = _pair(18, 'SimpleLinear', 'CrossEntropyLoss', K=K, dev=dev)
    ref_prev, mod_prev, _ = _pair(18, 'SimpleLinear', 'CrossEntropyLoss', K=K, dev=dev, seed=9)
    for m in (ref, mod, ref_prev, mod_prev):
        m.test_cfg['average_clips'] = 'score'
    imgs, labels = _clips(4, 8, 64, K)
    labels[0, 0], labels[1, 0] = 1, 9           # one old-class and one new-class sample at least
    ref.train(); mod.train(); ref_prev.eval(); mod_prev.eval()
    score = ref(imgs, return_loss=False)
    with torch.no_grad():
        prev_logits = ref_prev(imgs, return_loss=False)
    tgt = O.icarl_targets(labels, K, prev_logits, prevK)
    rloss = O.soft_target_ce(score, tgt)
    rloss.backward()
    oloss = bd.icarl_training_step(mod, dict(imgs=imgs.to(dev), label=labels.to(dev)), K, current_task=1, prev_model=mod_prev,
                                   previous_task_num_classes=prevK)
    oloss.backward()
    assert abs(oloss.item() - rloss.item()) <= 1e-4 * max(1.0, abs(rloss.item()))
    rp, op = dict(ref.named_parameters()), dict(mod.named_parameters())
    for name in ['backbone.conv1.conv.weight', 'backbone.layer4.1.conv2.conv.weight', 'cls_head.fc_cls.weight', 'cls_head.fc_cls.bias']:
        assert _rel_l2(op[name].grad, rp[name].grad) <= 2e-2, name
    # ActorCutMix keys in the batch: foreground-ratio soft labels (icarl.py:103-111), then the old-class rows
    g = torch.Generator().manual_seed(4)
    bg_label = torch.randint(-1, K, (4, 1), generator=g)
    bg_label[2, 0] = -1
    fg_ratio = torch.rand(4, 1, generator=g)
    tgt = O.icarl_targets(labels, K, prev_logits, prevK, bg_label, fg_ratio)
    with torch.no_grad():
        rloss2 = O.soft_target_ce(ref(imgs, return_loss=False), tgt)
    batch = dict(imgs=imgs.to(dev), label=labels.to(dev), background_label=bg_label.clone().to(dev), foreground_ratio=fg_ratio.to(dev))
    with torch.no_grad():
        oloss2 = bd.icarl_training_step(mod, batch, K, current_task=1, prev_model=mod_prev, previous_task_num_classes=prevK)
    assert abs(oloss2.item() - rloss2.item()) <= 1e-4 * max(1.0, abs(rloss2.item())) and abs(rloss2.item() - rloss.item()) > 1e-3
    assert int(batch['background_label'].min()) == 0                       # rewritten in place, as the reference does


def test_frontend_into_model(dev):
    """uint8 clips through the fused front-end == oracle bgmix_normalize + NCHW path."""
    import bdvcil_amd as bd
    ref, mod, _ = _pair(18, 'LocalSimilarityClassifier', 'LSCLoss', dev=dev)
    g = torch.Generator().manual_seed(3)
    B, T, S = 2, 8, 64
    fr = torch.randint(0, 256, (B, T, S, S, 3), generator=g, dtype=torch.uint8)
    bg = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    mix = torch.tensor([True, False])
    imgs = O.bgmix_normalize(fr, bg, mix, 0.5)
    ref.eval(); mod.eval()
    fe = bd.BackgroundMixFrontEnd(alpha=0.5)
    with torch.no_grad():
        r = ref.forward_test(imgs)
        o = mod.forward_test(fe(fr.to(dev), bg.to(dev), mix.to(dev))).cpu()
    assert (o - r).abs().max().item() <= 1e-3
    assert torch.equal(o.argmax(1), r.argmax(1))


def test_update_fc_and_state_dict_roundtrip(dev):
    ref, mod, cfg = _pair(18, 'LocalSimilarityClassifier', 'LSCLoss', K=5, dev=dev)
    import bdvcil_amd as bd
    old = mod.cls_head.fc_cls.weights.detach().clone()
    mod.update_fc(9)
    assert mod.cls_head.num_classes == 9 and tuple(mod.cls_head.fc_cls.weights.shape) == (9, 512)
    assert torch.equal(mod.cls_head.fc_cls.weights.detach()[:5], old) and mod.cls_head.fc_cls.weights.is_cuda
    prev = bd.build_model(copy.deepcopy(cfg)).to(dev)
    prev.update_fc(9)
    prev.load_state_dict(mod.state_dict())
    imgs, _ = _clips(1, 8, 64, 9)
    mod.eval(); prev.eval()
    with torch.no_grad():
        a, b = mod.forward_test(imgs.to(dev)), prev.forward_test(imgs.to(dev))
    assert torch.equal(a, b) and a.shape == (1, 9)


def test_multi_step_training_tracks_the_oracle(dev):
    """Six SGD steps (momentum, weight decay, clip 1.0, the reference's parameter groups) on a fixed batch: the HIP path
    and the CPU oracle start from the same weights and must follow the same loss curve, and the loss must go down
    (end-to-end sign/scale check of forward, backward and the fused optimizer)."""
    import bdvcil_amd as bd
    K_ = 7
    ref, mod, _ = _pair(18, 'LocalSimilarityClassifier', 'LSCLoss', K=K_, dev=dev, seed=3)
    imgs, labels = _clips(4, 8, 64, K_, seed=5)
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
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (hip_curve, ref_curve)


def test_stage_node_matches_block_nodes(dev):
    """A stage run as one autograd node (statistics of the block outputs taken in the next block's dgrad epilogue) gives
    the gradients of the block-by-block execution; a hook on an inner block switches the stage back automatically."""
    import bdvcil_amd as bd
    from bdvcil_amd import functional as Fn
    K_ = 9
    _, mod, _ = _pair(50, 'SimpleLinear', 'CrossEntropyLoss', K=K_, dev=dev, seed=2)
    imgs, labels = _clips(2, 8, 64, K_, seed=8)
    mod.train()

    def run():
        mod.zero_grad(set_to_none=True)
        loss = mod(imgs.to(dev), labels.to(dev))['loss_cls']
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}

    assert Fn.FUSE_STAGE
    l1, g1 = run()
    calls = []
    h = mod.backbone.layer2[1].register_forward_hook(lambda m, i, o: calls.append(tuple(o.shape)))
    l2, g2 = run()                       # layer2 now runs block by block (the hook must see the block output)
    h.remove()
    assert len(calls) == 1 and calls[0][1] == 512
    Fn.FUSE_STAGE = False
    try:
        l3, g3 = run()
    finally:
        Fn.FUSE_STAGE = True
    assert abs(l1 - l3) <= 1e-6 * max(1.0, abs(l3)) and abs(l2 - l3) <= 1e-6 * max(1.0, abs(l3))
    for n in g3:
        assert _rel_l2(g1[n], g3[n]) <= 2e-4, n
        assert _rel_l2(g2[n], g3[n]) <= 2e-4, n


def test_training_step_is_run_to_run_deterministic(dev):
    """No atomics anywhere on the path (split-K slabs, BN partial sums and loss reductions are summed in fixed order):
    two runs of the same step from the same state give bit-identical loss and gradients."""
    _, mod, _ = _pair(50, 'LocalSimilarityClassifier', 'LSCLoss', K=9, dev=dev, seed=4)
    imgs, labels = _clips(2, 8, 96, 9, seed=6)
    mod.train()
    x, y = imgs.to(dev), labels.to(dev)
    state = {k: v.clone() for k, v in mod.state_dict().items()}

    def run():
        mod.load_state_dict(state)
        mod.zero_grad(set_to_none=True)
        loss = mod(x, y)['loss_cls']
        loss.backward()
        return loss.detach().clone(), {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}

    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


def test_icarl_video_mix_step(dev):
    """ICARLVideoMix.training_step: tube-mix of the clips and targets, then the iCaRL soft-target step; the draws come
    from the same generators in the same order as the CPU restatement (parity unpinned: see oracle.tubemix)."""
    import random
    import bdvcil_amd as bd
    K, prevK = 9, 4
    ref, mod, _ = _pair(18, 'SimpleLinear', 'CrossEntropyLoss', K=K, dev=dev)
    ref_prev, mod_prev, _ = _pair(18, 'SimpleLinear', 'CrossEntropyLoss', K=K, dev=dev, seed=9)
    for m in (ref, mod, ref_prev, mod_prev):
        m.test_cfg['average_clips'] = 'score'
    ref.train(); mod.train(); ref_prev.eval(); mod_prev.eval()
    imgs, labels = _clips(4, 8, 64, K)
    labels[0, 0], labels[3, 0] = 2, 7
    mixed_steps = 0
    for seed in range(4):
        def seeds():
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        seeds()
        x = imgs.clone()
        xm, tgt = O.tubemix(x, F.one_hot(labels.view(-1), K).float(), (0.8,), 0.6)
        mixed_steps += int(not torch.equal(xm, imgs))
        with torch.no_grad():
            score = ref(xm, return_loss=False)
            prev_logits = ref_prev(xm, return_loss=False)
        old = (labels.view(-1) < prevK).nonzero().squeeze(1)
        tgt = tgt.clone()
        tgt[old] = torch.softmax(prev_logits[old], dim=1)
        rloss = O.soft_target_ce(score, tgt)
        seeds()
        xg = imgs.clone().to(dev)
        with torch.no_grad():
            oloss = bd.icarl_video_mix_training_step(mod, dict(imgs=xg, label=labels.to(dev)), K, 0.6, 0.8, current_task=1,
                                                     prev_model=mod_prev, previous_task_num_classes=prevK)
        assert torch.equal(xg.cpu(), xm)                                   # the same box from the same permuted samples
        assert abs(oloss.item() - rloss.item()) <= 1e-4 * max(1.0, abs(rloss.item())), seed
    assert 0 < mixed_steps < 4
    with pytest.raises(ValueError):
        bd.tubemix_draw(4, 8, 8, (0.8,), -0.1)


def test_full_size_step_properties(dev):
    """BASELINE config 2 at its full size (TSM-R50, 32 x 8 x 3 x 224 x 224, 101 classes), where the CPU oracle would take
    minutes: properties that hold for any correct implementation.

    * eval mode (running statistics): clips are independent, so the logits of the first four clips of the batch equal a
      batch-of-four run (the tile schedule differs, hence a tolerance), with identical labels;
    * train step with mean cross-entropy: the logit gradients of every sample sum to zero, so the classifier bias gradient
      sums to zero; every gradient is finite; the step is reproducible bit for bit;
    * linearity of backward: doubling the loss doubles every gradient (exactly: powers of two)."""
    import bdvcil_amd as bd
    torch.manual_seed(0)
    cfg = O.r50_cfg(num_classes=101, depth=50, head='SimpleLinear', loss='CrossEntropyLoss', dropout_ratio=0.0)
    mod = bd.build_model(copy.deepcopy(cfg)).to(dev)
    imgs, labels = _clips(32, 8, 224, 101)
    imgs, labels = imgs.to(dev), labels.to(dev)
    mod.eval()
    with torch.no_grad():
        full = mod(imgs, return_loss=False)
        part = mod(imgs[:4].contiguous(), return_loss=False)
    assert full.shape == (32, 101)
    assert (full[:4] - part).abs().max().item() <= 1e-4 * max(1.0, full.abs().max().item())
    assert torch.equal(full[:4].argmax(1), part.argmax(1))
    mod.train()

    def grads(scale):
        for p in mod.parameters():
            p.grad = None
        bn = {k: v.clone() for k, v in mod.state_dict().items() if 'running' in k or 'num_batches' in k}
        out = mod(imgs, labels)
        (out['loss_cls'] * scale).backward()
        mod.load_state_dict(bn, strict=False)                      # same running statistics for the next call
        return float(out['loss_cls'].detach()), {n: p.grad.clone() for n, p in mod.named_parameters()}
    l1, g1 = grads(1.0)
    l1b, g1b = grads(1.0)
    l2, g2 = grads(2.0)
    assert l1 == l1b == l2 and np.isfinite(l1)
    bias = g1['cls_head.fc_cls.fc.bias'] if 'cls_head.fc_cls.fc.bias' in g1 else g1['cls_head.fc_cls.bias']
    assert abs(float(bias.sum())) <= 1e-5 and float(bias.abs().max()) > 1e-4
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
        assert torch.equal(g1[n], g1b[n]), n                        # run-to-run deterministic
        assert torch.equal(g2[n], g1[n] * 2), n                     # backward is linear in the loss gradient
