"""CPU oracle (pure PyTorch, fp32 / fp64) for the I3D-ResNet50 path of BASELINE config 4.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

PARITY UNPINNED.  The reference only holds the config (configs/_base_/models/i3d_r50.py:1-27); ``Recognizer3D``, ``ResNet3d``,
``Bottleneck3d`` and ``I3DHead`` live in un-vendored mmaction2 0.24.x and no fixture, test or checkpoint of the reference
covers them.  This file restates the published mmaction2 module for exactly the settings that config selects:

* ``conv1``: Conv3d(3, 64, (5,7,7), stride (2,2,2), padding (2,3,3), bias=False) + BatchNorm3d + ReLU
  (``conv1_kernel=(5,7,7)``, ``conv1_stride_t=2``, :10-11); ``maxpool``: MaxPool3d((1,3,3), stride (2,2,2), padding (0,1,1))
  (``pool1_stride_t=2``, :12); ``pool2``: MaxPool3d((2,1,1), stride (2,1,1)) after layer1 (``with_pool2`` default True).
* stages 3/4/6/3 ``Bottleneck3d`` with planes 64/128/256/512, spatial strides 1/2/2/2, temporal strides 1, ``style='pytorch'``
  (stride on conv2), ``inflate_style='3x1x1'``: an inflated block has conv1 = (3,1,1) padding (1,0,0), otherwise (1,1,1);
  conv2 = (1,3,3) padding (0,1,1); conv3 = (1,1,1) without activation; downsample = (1,1,1) conv with the spatial stride + BN.
  ``inflate=((1,1,1),(1,0,1,0),(1,0,1,0,1,0),(0,1,0))`` (:15) marks the inflated blocks.
* initialisation without a checkpoint: kaiming-normal (fan_out, relu) convolutions, BatchNorm weight 1 / bias 0.
* ``I3DHead``: AdaptiveAvgPool3d(1) -> Dropout -> Linear(2048, K) with normal(0, init_std) weights (:17-23);
  ``Recognizer3D``: clips folded into the batch, ``average_clips`` over them at test time.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

I3D_INFLATE = ((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0))


class ConvModule3d(nn.Module):
    def __init__(self, cin, cout, k, stride=(1, 1, 1), padding=(0, 0, 0), act=True):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm3d(cout, eps=1e-5, momentum=0.1)
        self.with_act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.with_act else x


class Bottleneck3d(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, spatial_stride=1, inflate=True, downsample=None):
        super().__init__()
        if inflate:
            self.conv1 = ConvModule3d(inplanes, planes, (3, 1, 1), (1, 1, 1), (1, 0, 0))
        else:
            self.conv1 = ConvModule3d(inplanes, planes, (1, 1, 1))
        self.conv2 = ConvModule3d(planes, planes, (1, 3, 3), (1, spatial_stride, spatial_stride), (0, 1, 1))
        self.conv3 = ConvModule3d(planes, planes * 4, (1, 1, 1), act=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(self.conv3(self.conv2(self.conv1(x))) + identity)


class ResNet3d(nn.Module):
    def __init__(self, inflate=I3D_INFLATE, norm_eval=False, **_unused):
        super().__init__()
        self.norm_eval = norm_eval
        self.conv1 = ConvModule3d(3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3))
        self.maxpool = nn.MaxPool3d((1, 3, 3), stride=(2, 2, 2), padding=(0, 1, 1))
        self.pool2 = nn.MaxPool3d((2, 1, 1), stride=(2, 1, 1))
        inplanes = 64
        for i, (n, planes) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if b == 0 and (s != 1 or inplanes != planes * 4):
                    down = ConvModule3d(inplanes, planes * 4, (1, 1, 1), (1, s, s), act=False)
                blocks.append(Bottleneck3d(inplanes, planes, s, bool(inflate[i][b]), down))
                inplanes = planes * 4
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        x = self.pool2(self.layer1(x))
        return self.layer4(self.layer3(self.layer2(x)))


class I3DHead(nn.Module):
    def __init__(self, num_classes, in_channels=2048, dropout_ratio=0.5, init_std=0.01):
        super().__init__()
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio else None
        self.fc_cls = nn.Linear(in_channels, num_classes)
        nn.init.normal_(self.fc_cls.weight, 0, init_std)
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        x = F.adaptive_avg_pool3d(x, 1).flatten(1)
        if self.dropout is not None:
            x = self.dropout(x)
        return self.fc_cls(x)


class Recognizer3D(nn.Module):
    def __init__(self, num_classes, dropout_ratio=0.5, average_clips='prob'):
        super().__init__()
        self.backbone = ResNet3d()
        self.cls_head = I3DHead(num_classes, 2048, dropout_ratio)
        self.test_cfg = dict(average_clips=average_clips)

    def forward(self, imgs, label=None, return_loss=True):
        b, clips = imgs.shape[:2]
        score = self.cls_head(self.backbone(imgs.reshape((-1,) + imgs.shape[2:])))
        if return_loss:
            gt = label.squeeze()
            gt = gt.unsqueeze(0) if gt.dim() == 0 else gt
            pred = score.detach()
            top = pred.topk(min(5, pred.shape[1]), dim=1).indices
            return dict(loss_cls=F.cross_entropy(score, gt), top1_acc=(top[:, 0] == gt).float().mean(),
                        top5_acc=(top == gt[:, None]).any(1).float().mean())
        mode = self.test_cfg.get('average_clips')
        score = score.view(b, clips, -1)
        if mode == 'prob':
            return F.softmax(score, dim=2).mean(1)
        return score.mean(1) if mode == 'score' else score.view(b * clips, -1)


def i3d_conv_macs(T=32, S=224):
    """Multiply-accumulates of all convolutions for one clip of T frames at S x S (this file's restatement)."""
    net = ResNet3d()
    macs = [0]

    def hook(m, i, o):
        macs[0] += o.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2]
    for m in net.modules():
        if isinstance(m, nn.Conv3d):
            m.register_forward_hook(hook)
    net.eval()
    with torch.no_grad():
        net(torch.zeros(1, 3, T, S, S))
    return macs[0]
