"""CPU oracle (pure PyTorch, fp32) for the TSM class-incremental hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Nothing in the product
package imports this file; it is the checker the HIP path is compared against.

Parity pinning status
---------------------
* PINNED by golden vectors generated from the reference's own importable files
  (``tests/golden/make_golden.py`` -> ``tests/golden/head_loss_golden.npz``):
  ``lsc_forward`` / ``LSC`` (libs/models/cil_heads/cosine_linear.py:27-43),
  ``IncrementalNet`` (libs/models/cil_heads/inc_net.py:23-37),
  ``lsc_loss`` (libs/losses/lsc_loss.py:30-58); ``acm_smooth_ce`` -- and with it the foreground-ratio soft labels of
  ``icarl_targets`` -- by ``tests/golden/acm_golden.npz`` (libs/losses/acm_smooth_ce.py:13-30).
* PARITY UNPINNED: ``tubemix`` (libs/cil/icarl_video_mix.py:48-81 uses ``np.int``, which this image's numpy no longer
  has, and sits behind Lightning imports).
* PARITY UNPINNED: everything that lives in un-vendored mmaction2 0.24.x / mmcv 1.x
  (ResNet, TemporalShift, TSMHead, Recognizer2D, BaseHead.loss, Normalize).  Those
  packages are absent from /root/reference and from this image; the restatement
  follows SURVEY.md Appendix A and the reference's call sites
  (libs/models/base.py:10-28, libs/models/cil_heads/tsm.py:21-64,
  configs/ucf101/bgmix_plus_randAug/bgmix_seed_1000_inc_10_stages_bgmix_plus_randAug.py:57-83).
  The reference ships no tests or fixtures for them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------
# temporal shift  (UPSTREAM mmaction TemporalShift.shift)
# ---------------------------------------------------------------------------------------


def temporal_shift(x: torch.Tensor, num_segments: int, shift_div: int) -> torch.Tensor:
    """x: (N=B*T, C, H, W).  out[:, t, :f] = x[:, t+1, :f]; out[:, t, f:2f] = x[:, t-1, f:2f];
    zero fill at the clip ends; the remaining channels are copied."""
    n, c, h, w = x.shape
    t = num_segments
    v = x.view(n // t, t, c, h * w)
    fold = c // shift_div
    out = torch.zeros_like(v)
    out[:, :-1, :fold] = v[:, 1:, :fold]
    out[:, 1:, fold:2 * fold] = v[:, :-1, fold:2 * fold]
    out[:, :, 2 * fold:] = v[:, :, 2 * fold:]
    return out.view(n, c, h, w)


class TemporalShift(nn.Module):
    """Wrapper giving the ``...conv1.conv.net.weight`` checkpoint key."""

    def __init__(self, net: nn.Module, num_segments: int, shift_div: int):
        super().__init__()
        self.net = net
        self.num_segments = num_segments
        self.shift_div = shift_div

    def forward(self, x):
        return self.net(temporal_shift(x, self.num_segments, self.shift_div))


# ---------------------------------------------------------------------------------------
# ResNet / ResNetTSM  (UPSTREAM mmaction ResNet + ResNetTSM)
# ---------------------------------------------------------------------------------------


class ConvModule(nn.Module):
    """conv(bias=False) -> BatchNorm2d(eps 1e-5, momentum 0.1) -> optional ReLU."""

    def __init__(self, cin, cout, k, stride=1, padding=0, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5, momentum=0.1)
        self.with_act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.with_act else x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvModule(inplanes, planes, 3, stride, 1, act=True)
        self.conv2 = ConvModule(planes, planes, 3, 1, 1, act=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.conv2(self.conv1(x))
        return F.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = ConvModule(inplanes, planes, 1, 1, 0, act=True)
        self.conv2 = ConvModule(planes, planes, 3, stride, 1, act=True)
        self.conv3 = ConvModule(planes, planes * 4, 1, 1, 0, act=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.conv3(self.conv2(self.conv1(x)))
        return F.relu(out + identity)


ARCH = {
    18: (BasicBlock, (2, 2, 2, 2)),
    34: (BasicBlock, (3, 4, 6, 3)),
    50: (Bottleneck, (3, 4, 6, 3)),
}


class ResNetTSM(nn.Module):
    def __init__(self, depth=50, num_segments=8, shift_div=8, is_shift=True,
                 norm_eval=False, pretrained=None, **_unused):
        super().__init__()
        block, counts = ARCH[depth]
        self.depth = depth
        self.num_segments = num_segments
        self.shift_div = shift_div
        self.norm_eval = norm_eval
        self.conv1 = ConvModule(3, 64, 7, 2, 3, act=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (n, planes) in enumerate(zip(counts, (64, 128, 256, 512))):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if b == 0 and (s != 1 or inplanes != planes * block.expansion):
                    down = ConvModule(inplanes, planes * block.expansion, 1, s, 0, act=False)
                blocks.append(block(inplanes, planes, s, down))
                inplanes = planes * block.expansion
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        self.feat_dim = inplanes
        self.init_weights()
        if is_shift:
            self.make_temporal_shift()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def make_temporal_shift(self):
        for li in range(1, 5):
            for blk in getattr(self, f'layer{li}'):
                blk.conv1.conv = TemporalShift(blk.conv1.conv, self.num_segments, self.shift_div)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        return self.layer4(x)

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


# ---------------------------------------------------------------------------------------
# incremental classifiers (reference: libs/models/cil_heads/cosine_linear.py, inc_net.py)
# ---------------------------------------------------------------------------------------


def lsc_forward(x: torch.Tensor, weights: torch.Tensor, out_features: int, nb_proxies: int):
    """cosine_linear.py:27-43.  x (N, D); weights (K, P*D) -> (N, K)."""
    d = x.shape[1]
    w = weights.reshape(out_features * nb_proxies, d)
    sims = F.cosine_similarity(x[:, None, :], w[None, :, :], dim=2)          # (N, K*P)
    per_class = sims.reshape(-1, out_features, nb_proxies)
    attn = torch.softmax(per_class, dim=2)
    return (attn * per_class).sum(dim=2)


class LSC(nn.Module):
    def __init__(self, in_features, out_features, nb_proxies=3):
        super().__init__()
        self.in_features, self.out_features, self.nb_proxies = in_features, out_features, nb_proxies
        self.weights = nn.Parameter(torch.empty(out_features, nb_proxies * in_features))
        nn.init.kaiming_normal_(self.weights, nonlinearity='linear')

    def forward(self, x):
        return lsc_forward(x, self.weights, self.out_features, self.nb_proxies)

    def update_fc(self, nb_classes):
        """cosine_linear.py:45-50: new tensor, kaiming_normal_, old rows copied."""
        new = torch.empty(nb_classes, self.nb_proxies * self.in_features).type_as(self.weights.data)
        nn.init.kaiming_normal_(new, nonlinearity='linear')
        new[:self.out_features] = self.weights.data
        self.weights = nn.Parameter(new)
        self.out_features = nb_classes


class IncrementalNet(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.kaiming_uniform_(self.weight, nonlinearity='linear')

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)

    def update_fc(self, nb_classes):
        """inc_net.py:23-34."""
        w = torch.empty(nb_classes, self.in_features).type_as(self.weight.data)
        nn.init.kaiming_normal_(w, nonlinearity='linear')
        w[:self.out_features] = self.weight.data
        b = torch.zeros(nb_classes).type_as(self.bias.data)
        b[:self.out_features] = self.bias.data
        self.weight, self.bias = nn.Parameter(w), nn.Parameter(b)
        self.out_features = nb_classes


# ---------------------------------------------------------------------------------------
# losses (reference: libs/losses/lsc_loss.py, libs/cil/icarl.py:100-125)
# ---------------------------------------------------------------------------------------


def lsc_loss(sim: torch.Tensor, targets: torch.Tensor, eta: torch.Tensor, margin: float = 0.6,
             hinge: bool = True) -> torch.Tensor:
    """lsc_loss.py:36-56 (exclude_pos_denominator=True).  NOTE Appendix C.1: the positive
    slot is zeroed, not removed, so exp(0)=1 stays in the denominator."""
    s = eta * (sim - margin)
    s = s - s.max(dim=1, keepdim=True)[0]
    idx = torch.arange(s.shape[0])
    num = s[idx, targets]
    den = s.clone()
    den[idx, targets] = 0.0
    losses = -(num - torch.log(torch.exp(den).sum(-1)))
    if hinge:
        losses = torch.clamp(losses, min=0.0)
    return losses.mean()


class LSCLoss(nn.Module):
    def __init__(self, eta=1.0, margin=0.6, learnable_eta=True, exclude_pos_denominator=True,
                 hinge_proxynca=True, class_weights=None):
        super().__init__()
        assert exclude_pos_denominator and class_weights is None
        self.margin, self.hinge_proxynca, self.learnable_eta = margin, hinge_proxynca, learnable_eta
        self.eta = nn.Parameter(torch.tensor([float(eta)]), requires_grad=learnable_eta)

    def forward(self, sim, targets, **kwargs):
        return lsc_loss(sim, targets, self.eta, self.margin, self.hinge_proxynca)


class CrossEntropyLoss(nn.Module):
    """Plain mean CE on integer labels; extra kwargs ignored (BASELINE config 2: "CE loss only")."""

    def forward(self, cls_score, labels, **kwargs):
        return F.cross_entropy(cls_score, labels)


def soft_target_ce(cls_score: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """icarl.py:123-125: mean_b( -sum_k tgt * log_softmax(score) )."""
    return (-(targets * F.log_softmax(cls_score, dim=1)).sum(dim=1)).mean(dim=0)


def icarl_targets(labels: torch.Tensor, num_classes: int, prev_logits: Optional[torch.Tensor],
                  prev_num_classes: int, background_label: Optional[torch.Tensor] = None,
                  foreground_ratio: Optional[torch.Tensor] = None) -> torch.Tensor:
    """icarl.py:101-120: one-hot (mixed with the background label by lambda = 1 - (1 - foreground_ratio)^4 when the
    ActorCutMix keys are present, :103-111), rows of old-class samples replaced by softmax(prev logits)."""
    tgt = F.one_hot(labels.view(-1), num_classes).float()
    if foreground_ratio is not None:
        bg = torch.squeeze(background_label, dim=1).clone()
        bg[bg == -1] = 0
        bg = F.one_hot(bg, num_classes).float()
        lam = 1 - (1 - foreground_ratio) ** 4
        lam = lam.view(lam.size(0), 1)
        tgt = (tgt * lam + (1 - lam) * bg).float()
    if prev_logits is not None:
        old = (labels.view(-1) < prev_num_classes).nonzero().squeeze(1)
        if old.numel():
            tgt[old] = torch.softmax(prev_logits[old], dim=1)
    return tgt


def tubemix(x: torch.Tensor, y: torch.Tensor, alpha, prob: float):
    """libs/cil/icarl_video_mix.py:48-81 (tubemix + rand_bbox) on a (B, T, 3, H, W) clip batch and (B, K) one-hot targets,
    drawing from ``random`` / ``torch`` / ``np.random`` in the reference's order.  ``np.int`` is spelled ``int``.
    Parity unpinned: the reference function cannot run on this image's numpy (np.int was removed)."""
    import random

    import numpy as np
    if prob < 0:
        raise ValueError('prob must be a positive value')
    k = random.random()
    if k > 1 - prob:
        batch_idx = torch.randperm(x.size(0))
        lam = np.random.beta(alpha, alpha)
        size = x[:, :, 0, :, :].size()
        W_, H_ = size[2], size[3]
        cut_rat = np.sqrt(1. - lam)
        cut_w, cut_h = int(np.asarray(W_ * cut_rat).reshape(-1)[0]), int(np.asarray(H_ * cut_rat).reshape(-1)[0])
        cx, cy = np.random.randint(W_), np.random.randint(H_)
        bbx1, bby1 = np.clip(cx - cut_w // 2, 0, W_), np.clip(cy - cut_h // 2, 0, H_)
        bbx2, bby2 = np.clip(cx + cut_w // 2, 0, W_), np.clip(cy + cut_h // 2, 0, H_)
        x[:, :, :, bbx1:bbx2, bby1:bby2] = x[batch_idx, :, :, bbx1:bbx2, bby1:bby2]
        lam = 1 - ((bbx2 - bbx1) * (bby2 - bby1) / (x.size()[-1] * x.size()[-2]))
        return x, y * lam + y[batch_idx] * (1 - lam)
    return x, y


def acm_smooth_ce(cls_score: torch.Tensor, labels: torch.Tensor, background_label: torch.Tensor,
                  foreground_ratio: torch.Tensor, num_classes: int, alpha: float = 4.0) -> torch.Tensor:
    """libs/losses/acm_smooth_ce.py:13-30 (sign as in the reference: no negation; background label -1 -> 0)."""
    action = F.one_hot(labels, num_classes=num_classes)
    bg = torch.squeeze(background_label, dim=1).clone()
    bg[bg == -1] = 0
    bg = F.one_hot(bg, num_classes=num_classes)
    lam = 1 - (1 - foreground_ratio) ** alpha
    y = action * lam + (1 - lam) * bg
    return torch.mean(torch.sum(y * F.log_softmax(cls_score, dim=1), dim=1), dim=0)


def top_k_hits(scores: torch.Tensor, labels: torch.Tensor, k: int) -> float:
    """UPSTREAM mmaction top_k_accuracy (hit iff the label is among the k best scores)."""
    lab = scores.gather(1, labels.view(-1, 1))
    rank = (scores > lab).sum(dim=1)
    return float((rank < k).float().mean())


# ---------------------------------------------------------------------------------------
# head + recognizer (UPSTREAM TSMHead/Recognizer2D; reference tsm.py:21-64, base.py:10-28)
# ---------------------------------------------------------------------------------------

INC_LAYERS = {'SimpleLinear': IncrementalNet, 'LocalSimilarityClassifier': LSC}
LOSSES = {'LSCLoss': LSCLoss, 'CrossEntropyLoss': CrossEntropyLoss}


class AvgConsensus(nn.Module):
    def __init__(self, dim=1):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.mean(dim=self.dim, keepdim=True)


class IncrementalTSMHead(nn.Module):
    def __init__(self, num_classes, in_channels, inc_head_config=None, num_segments=8,
                 loss_cls=None, spatial_type='avg', consensus=None, dropout_ratio=0.8,
                 init_std=0.001, is_shift=True, temporal_pool=False, **_unused):
        super().__init__()
        self.num_classes, self.in_channels, self.num_segments = num_classes, in_channels, num_segments
        cfg = dict(inc_head_config or dict(type='LocalSimilarityClassifier'))
        cfg['in_features'] = in_channels
        lcfg = dict(loss_cls or dict(type='CrossEntropyLoss'))
        self.loss_cls = LOSSES[lcfg.pop('type')](**lcfg)
        self.consensus = AvgConsensus(**{k: v for k, v in (consensus or {}).items() if k != 'type'})
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio else None
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        layer = INC_LAYERS[cfg.pop('type')]
        self.fc_cls = layer(**cfg)
        self.fc_cls.update_fc(num_classes)      # tsm.py:51-56

    def update_fc(self, nb_classes):
        self.fc_cls.update_fc(nb_classes)
        self.num_classes = nb_classes

    def forward(self, x, num_segs=None):
        x = torch.flatten(self.avg_pool(x), 1)
        if self.dropout is not None:
            x = self.dropout(x)
        s = self.fc_cls(x)
        s = s.view((-1, self.num_segments) + s.shape[1:])
        return self.consensus(s).squeeze(1)

    def loss(self, cls_score, labels, **kwargs):
        out = {}
        if labels.shape == torch.Size([]):
            labels = labels.unsqueeze(0)
        out['top1_acc'] = torch.tensor(top_k_hits(cls_score.detach(), labels, 1))
        out['top5_acc'] = torch.tensor(top_k_hits(cls_score.detach(), labels, 5))
        out['loss_cls'] = self.loss_cls(cls_score, labels, **kwargs)
        return out


class CILRecognizer2D(nn.Module):
    def __init__(self, backbone, cls_head, train_cfg=None, test_cfg=None, **_unused):
        super().__init__()
        bcfg = {k: v for k, v in backbone.items() if k != 'type'}
        hcfg = {k: v for k, v in cls_head.items() if k != 'type'}
        self.backbone = ResNetTSM(**bcfg)
        self.cls_head = IncrementalTSMHead(**hcfg)
        self.train_cfg, self.test_cfg = train_cfg, dict(test_cfg or {})

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs)

    def _scores(self, imgs):
        x = imgs.reshape((-1,) + imgs.shape[2:])
        return self.cls_head(self.backbone(x))

    def forward_train(self, imgs, labels, **kwargs):
        cls_score = self._scores(imgs)
        return self.cls_head.loss(cls_score, labels.squeeze(),
                                  num_classes=self.cls_head.num_classes, **kwargs)

    def forward_test(self, imgs):
        batches = imgs.shape[0]
        cls_score = self._scores(imgs)
        n = cls_score.shape[0] // batches
        s = cls_score.view(batches, n, -1)
        mode = self.test_cfg.get('average_clips')
        if mode == 'prob':
            return torch.softmax(s, dim=2).mean(dim=1)
        if mode == 'score':
            return s.mean(dim=1)
        return cls_score

    def update_fc(self, nb_classes):
        self.cls_head.update_fc(nb_classes)


def build_model(cfg: dict) -> CILRecognizer2D:
    cfg = {k: v for k, v in cfg.items() if k != 'type'}
    return CILRecognizer2D(**cfg)


# ---------------------------------------------------------------------------------------
# background mix + normalize front-end (reference: libs/loader/comix_loader.py:72-75,138-145)
# ---------------------------------------------------------------------------------------

IMG_MEAN = (123.675, 116.28, 103.53)
IMG_STD = (58.395, 57.12, 57.375)


def bgmix_normalize(frames_u8: torch.Tensor, bg_u8: torch.Tensor, mix_mask: torch.Tensor,
                    alpha: float = 0.5, mean: Sequence[float] = IMG_MEAN,
                    std: Sequence[float] = IMG_STD) -> torch.Tensor:
    """frames_u8 (B,T,H,W,3) uint8 RGB, bg_u8 (B,H,W,3) uint8, mix_mask (B,) bool.
    Returns (B,T,3,H,W) fp32.  Frames: mmcv imnormalize ``(x-mean)*(1/std)``; background:
    torchvision Normalize ``(x-mean)/std``; blend ``x*(1-a)+bg*a`` on normalised tensors
    (comix_loader.py:142), applied only where ``mix_mask`` is set (comix_loader.py:110-116)."""
    m = torch.tensor(mean, dtype=torch.float32)
    s = torch.tensor(std, dtype=torch.float32)
    x = (frames_u8.float() - m) * (1.0 / s)                    # (B,T,H,W,3)
    b = (bg_u8.float() - m) / s                                # (B,H,W,3)
    blend = x * (1 - alpha) + b[:, None] * alpha
    out = torch.where(mix_mask.view(-1, 1, 1, 1, 1), blend, x)
    return out.permute(0, 1, 4, 2, 3).contiguous()


def bg_resize_crop(bg_u8: torch.Tensor, resize: int, crop: int, top: int, left: int, antialias: bool = False) -> torch.Tensor:
    """One background image through ``Resize(resize) -> RandomCrop(crop)`` of BackgroundMixDataset.bg_pipeline
    (libs/loader/comix_loader.py:72-73) at the given crop offsets.  bg_u8 (Hs,Ws,3) uint8 -> (crop,crop,3) fp32 in [0,255].
    UPSTREAM torchvision (not importable here, version unpinned by the reference: PARITY UNPINNED): ``Resize`` with an int
    makes the smaller edge ``resize`` and the other ``int(resize * long / short)``; on a float tensor image (the reference
    passes ``read_image(...).float()``, comix_loader.py:128,134) it is ``F.interpolate(mode='bilinear', align_corners=False)``
    without rounding; ``antialias`` only matters when the image shrinks (torchvision's default changed over versions)."""
    import torch.nn.functional as F
    h, w = bg_u8.shape[:2]
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = resize, int(resize * long / short)
    nh, nw = (new_long, new_short) if w <= h else (new_short, new_long)
    img = bg_u8.permute(2, 0, 1).float().unsqueeze(0)                       # read_image gives (3,H,W); .float()
    out = F.interpolate(img, size=(nh, nw), mode='bilinear', align_corners=False, antialias=antialias)[0]
    return out[:, top:top + crop, left:left + crop].permute(1, 2, 0).contiguous()


# ---------------------------------------------------------------------------------------
# optimizer param groups (reference: libs/models/cil_heads/tsm.py:211-303) + step
# ---------------------------------------------------------------------------------------


def param_groups(model: nn.Module, lr: float, wd: float, fc_lr_scale_factor: float = 5.0) -> List[dict]:
    first_w, first_b, normal_w, normal_b, bn, lr5_w, lr10_b = [], [], [], [], [], [], []
    conv_cnt = 0
    for m in model.modules():
        if isinstance(m, nn.modules.conv._ConvNd):
            ps = list(m.parameters())
            conv_cnt += 1
            (first_w if conv_cnt == 1 else normal_w).append(ps[0])
            if len(ps) == 2:
                (first_b if conv_cnt == 1 else normal_b).append(ps[1])
        elif isinstance(m, nn.Linear):
            ps = list(m.parameters())
            normal_w.append(ps[0])
            if len(ps) == 2:
                normal_b.append(ps[1])
        elif isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):
            bn.extend(p for p in m.parameters() if p.requires_grad)
        elif isinstance(m, LSC):
            lr5_w.append(m.weights)
        elif isinstance(m, LSCLoss):
            if m.learnable_eta:
                lr5_w.append(m.eta)
        elif isinstance(m, IncrementalNet):
            lr5_w.append(m.weight)
            lr10_b.append(m.bias)
        elif len(m._modules) == 0 and len(list(m.parameters())) > 0:
            raise ValueError(f'New atomic module type: {type(m)}. Need to give it a learning policy')
    return [
        dict(params=first_w, lr=lr, weight_decay=wd),
        dict(params=first_b, lr=lr * 2, weight_decay=0),
        dict(params=normal_w, lr=lr, weight_decay=wd),
        dict(params=normal_b, lr=lr * 2, weight_decay=0),
        dict(params=bn, lr=lr, weight_decay=0),
        dict(params=lr5_w, lr=lr * fc_lr_scale_factor, weight_decay=wd),
        dict(params=lr10_b, lr=lr * fc_lr_scale_factor * 2, weight_decay=0),
    ]


def build_sgd(model, lr=0.01, momentum=0.9, weight_decay=1e-4, fc_lr_scale_factor=5.0):
    groups = [g for g in param_groups(model, lr, weight_decay, fc_lr_scale_factor) if g['params']]
    return torch.optim.SGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)


# ---------------------------------------------------------------------------------------
# CIL training-step arithmetic (reference: libs/cil/cil.py:512-556)
# ---------------------------------------------------------------------------------------


class FeatureTap:
    """Forward-hook tap by dotted name (reference: libs/module_hooks/output_hook.py)."""

    def __init__(self, model: nn.Module, names: Sequence[str]):
        self.out: Dict[str, torch.Tensor] = {}
        self.handles = []
        for name in names:
            mod = model
            for part in name.split('.'):
                mod = getattr(mod, part)
            self.handles.append(mod.register_forward_hook(self._make(name)))

    def _make(self, name):
        def hook(_m, _i, o):
            self.out[name] = o
        return hook


def kd_training_step(cur: CILRecognizer2D, prev: Optional[CILRecognizer2D], cur_tap: FeatureTap,
                     prev_tap: Optional[FeatureTap], imgs, labels, kd_names: Sequence[str],
                     kd_weights: Sequence[float], scale_factor: float, use_kd: bool, kd_exemplar_only: bool = False,
                     previous_task_num_classes: int = 0) -> Dict[str, torch.Tensor]:
    """cil.py:512-556."""
    losses = cur(imgs, labels, batch_data=None)
    if use_kd and prev is not None:
        prev.eval()
        with torch.no_grad():
            prev.forward_test(imgs)
        total = 0.0
        for name, w in zip(kd_names, kd_weights):
            c, p = cur_tap.out[name], prev_tap.out[name].detach()
            if kd_exemplar_only:                                                   # cil.py:529-536
                indices = (labels.view(-1) < previous_task_num_classes).nonzero().squeeze()
                mse = F.mse_loss(c[indices], p[indices]) if indices.nelement() else 0
            else:
                mse = F.mse_loss(c, p)
            losses[name] = mse
            total = total + scale_factor * w * mse
        losses['kd_loss'] = total
    else:
        losses['kd_loss'] = 0.0
    losses['loss'] = losses['kd_loss'] + losses['loss_cls']
    return losses


def r50_cfg(num_classes=101, depth=50, head='SimpleLinear', loss='CrossEntropyLoss',
            nb_proxies=1, dropout_ratio=0.5, num_segments=8):
    """Model dict in the reference's config shape (…bgmix_seed_1000_…:57-83)."""
    return dict(
        type='CILRecognizer2D',
        backbone=dict(type='ResNetTSM', pretrained=None, depth=depth, norm_eval=False,
                      num_segments=num_segments, shift_div=8),
        cls_head=dict(type='IncrementalTSMHead', num_classes=num_classes,
                      in_channels=2048 if depth == 50 else 512,
                      inc_head_config=(dict(type=head, out_features=num_classes, nb_proxies=nb_proxies)
                                       if head == 'LocalSimilarityClassifier'
                                       else dict(type=head, out_features=num_classes)),
                      num_segments=num_segments, loss_cls=dict(type=loss), spatial_type='avg',
                      consensus=dict(type='AvgConsensus', dim=1), dropout_ratio=dropout_ratio,
                      init_std=0.001, is_shift=True),
        train_cfg=None, test_cfg=dict(average_clips='prob'))
