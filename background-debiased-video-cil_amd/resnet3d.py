"""I3D-ResNet50 (UPSTREAM mmaction2 ``ResNet3d`` / ``Bottleneck3d`` / ``I3DHead`` / ``Recognizer3D`` as configured by
configs/_base_/models/i3d_r50.py:1-27) on the same HIP kernels as the TSM path -- SURVEY.md section 8(f) rank 4,
BASELINE.json config 4.  Parity unpinned: mmaction2 is not vendored and the reference holds no fixture for this model; the CPU
restatement used by the parity tests lives in the checker directory (i3d_*.py).

How the 3-D network maps onto the 2-D kernels (activations stay fp32 NHWC frames, ``[B*T][H][W][C]``):

* ``1 x k x k`` and ``1 x 1 x 1`` convolutions (conv2, conv3, downsample) are per-frame convolutions: N = B*T.
* the inflated ``3 x 1 x 1`` conv1 of a bottleneck runs as a ``3 x 1`` convolution on the ``[B][T][H*W][C]`` view of the same
  storage (``kernels.make_temporal_geom``: row padding 1, column padding 0): same implicit-GEMM kernels, no data movement.
* BatchNorm3d over (B, T, H, W) is BatchNorm over all frames: the 2-D kernels on M = B*T*H*W rows.
* the stem ``5 x 7 x 7`` / stride (2, 2, 2) convolution is five per-frame ``7 x 7`` / 2 stem convolutions over the frame
  subsets ``2t + dt - 2``, accumulated through the conv epilogue's residual input; ``pool1`` (1 x 3 x 3, stride 2 in time)
  is the spatial max-pool of every second frame; ``pool2`` (2 x 1 x 1) is ``bdv_maxpool_t2``.
* ``norm_eval=False``, ``inflate_style='3x1x1'``, ``non_local`` off, ``with_pool2=True``, temporal strides 1: the i3d_r50 config.
"""
from __future__ import annotations

from typing import List

import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .heads import AvgConsensus, HipDropout  # noqa: F401
from .registry import BACKBONES, HEADS, RECOGNIZERS, build_backbone, build_head, build_loss
from .resnet_tsm import ResStage, _ResBlock


def _channels_last_3d_(conv: nn.Conv3d):
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last_3d)
    return conv


class ConvModule3d(nn.Module):
    """Conv3d(bias=False) -> BatchNorm3d -> optional ReLU; children ``conv`` / ``bn`` as in mmcv's ConvModule."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=(1, 1, 1), padding=(0, 0, 0), act=True, frames=1):
        super().__init__()
        self.conv = _channels_last_3d_(nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False))
        self.bn = nn.BatchNorm3d(out_channels, eps=1e-5, momentum=0.1)
        self.with_activation = act
        self.frames = frames                     # frames per clip at this depth of the network (set by ResNet3d.forward)

    @property
    def raw_conv(self) -> nn.Conv3d:
        return self.conv

    def spec(self) -> Fn.UnitSpec:
        c = self.conv
        kt, kh, kw = c.kernel_size
        if c.stride[0] != 1:
            raise NotImplementedError('temporal stride inside a residual stage (i3d_r50 uses 1)')
        if kt > 1:
            if (kh, kw) != (1, 1) or c.padding != (kt // 2, 0, 0) or c.stride != (1, 1, 1):
                raise NotImplementedError(f'Conv3d {c.kernel_size}: only kt x 1 x 1 (inflate_style 3x1x1) is on the HIP path')
            return Fn.TemporalUnitSpec(c.in_channels, c.out_channels, kt, self.with_activation, self.frames)
        if kh != kw or c.stride[1] != c.stride[2] or c.padding[1] != c.padding[2] or c.padding[0] != 0:
            raise NotImplementedError(f'Conv3d {c.kernel_size} stride {c.stride} padding {c.padding}')
        return Fn.UnitSpec(c.in_channels, c.out_channels, kh, c.stride[1], c.padding[1], self.with_activation)


class Bottleneck3d(_ResBlock):
    """UPSTREAM Bottleneck3d, ``style='pytorch'``, ``inflate_style='3x1x1'``: conv1 3x1x1 (inflated) or 1x1x1, conv2 1x3x3 with
    the spatial stride, conv3 1x1x1; downsample 1x1x1 with the spatial stride."""
    expansion = 4
    main_names = ['conv1', 'conv2', 'conv3']

    def __init__(self, inplanes, planes, spatial_stride=1, inflate=True, downsample=None):
        super().__init__()
        if inflate:
            self.conv1 = ConvModule3d(inplanes, planes, (3, 1, 1), (1, 1, 1), (1, 0, 0), act=True)
        else:
            self.conv1 = ConvModule3d(inplanes, planes, (1, 1, 1), act=True)
        self.conv2 = ConvModule3d(planes, planes, (1, 3, 3), (1, spatial_stride, spatial_stride), (0, 1, 1), act=True)
        self.conv3 = ConvModule3d(planes, planes * 4, (1, 1, 1), act=False)
        self.downsample = downsample
        self._finalize()


class _Stem3dFn(torch.autograd.Function):
    """conv1 (5x7x7, stride 2,2,2) + BatchNorm3d + ReLU + pool1 (1x3x3, stride 2,2,2) of ResNet3d on NHWC4 frames.

    x4: (B*T, H, W, 4), T frames per clip.  Output: (B*T/4, H/4, W/4, 64) frames."""

    @staticmethod
    def forward(ctx, x4, weight, gamma, beta, bn, training, T):
        N, H, W, _ = x4.shape
        B = N // T
        Cout, _, kt, kh, kw = weight.shape
        To = (T + 2 * (kt // 2) - kt) // 2 + 1
        # One kernel over all kt x kh x kw taps (the stem kernels of the bf16-piece arithmetic take temporal taps: output frame n
        # reads the input frames 2 n + dt - kt // 2 of its clip) when the clip length fits (T = 2 To); otherwise kt accumulated 2-D
        # convolutions over gathered frames.
        fused = (T == 2 * To and K.FPROP_X3 and K.WGRAD_X3 and K.USE_PL_WGRAD and os.environ.get('BDVCIL_C4_X3', '1') != '0'
                 and os.environ.get('BDVCIL_STEM3D_FUSED', '1') != '0')
        taps = []
        if fused:
            g = K.make_geom(B * To, H, W, 4, Cout, kh, kw, 2, kh // 2, T=To, rt=kt, st_t=2)
            w4 = torch.zeros((Cout, kt, kh, kw, 4), dtype=torch.float32, device=x4.device)
            w4[..., :3] = weight.detach().permute(0, 2, 3, 4, 1)         # (Cout, kt, kh, kw, 3)
            y = K.conv_fprop(x4, w4.view(Cout, kt * kh, kw, 4), g)
            taps = [x4]
        else:
            g = K.make_geom(B * To, H, W, 4, Cout, kh, kw, 2, kh // 2)
            # per temporal tap dt: the frames 2t + dt - 2 of every clip (zero frames outside the clip), and the tap's 7x7 filter
            xv = x4.view(B, T, H, W, 4)
            w5 = weight.detach().permute(2, 0, 3, 4, 1)                  # (kt, Cout, kh, kw, 3)
            ws = []
            for dt in range(kt):
                idx = [2 * t + dt - kt // 2 for t in range(To)]
                sel = torch.zeros((B, To, H, W, 4), dtype=torch.float32, device=x4.device)
                ok = [k for k, i in enumerate(idx) if 0 <= i < T]
                if ok:
                    sel[:, ok[0]:ok[-1] + 1] = xv[:, idx[ok[0]]:idx[ok[-1]] + 1:2]
                taps.append(sel.view(B * To, H, W, 4))
                w4 = torch.zeros((Cout, kh, kw, 4), dtype=torch.float32, device=x4.device)
                w4[..., :3] = w5[dt]
                ws.append(w4)
            one = torch.ones(Cout, dtype=torch.float32, device=x4.device)
            zero = torch.zeros(Cout, dtype=torch.float32, device=x4.device)
            y = K.conv_fprop(taps[0], ws[0], g)
            for dt in range(1, kt):                                       # y += conv(tap dt): residual input of the folded epilogue
                y = K.conv_fprop(taps[dt], ws[dt], g, affine=(one, zero, y, False))
        save = training and any(ctx.needs_input_grad)
        if training:
            rm = bn.running_mean if bn.track_running_stats else None
            rv = bn.running_var if bn.track_running_stats else None
            mean, invstd, scale, shift = K.bn_train_stats(y, gamma, beta, bn.eps, bn.momentum, rm, rv)
        else:
            scale, shift = K.bn_eval_params(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
            mean = invstd = None
        if save:
            a, mask = K.bn_apply(y, scale, shift, None, True, want_mask=True)
        else:
            a, mask = K.bn_apply(y, scale, shift, None, True), None
        # pool1: temporal kernel 1, stride 2 -> the even frames; spatial 3x3 / 2
        Tp = (To - 1) // 2 + 1
        even = a.view(B, To, g.Ho, g.Wo, Cout)[:, ::2].contiguous().view(B * Tp, g.Ho, g.Wo, Cout)
        p, pidx = K.maxpool_fwd(even)
        if Fn.RELU_MASK_TAP is not None and mask is not None:      # tests: the stem's ReLU sign bits / pool1's arg-max codes
            Fn.RELU_MASK_TAP.append((tuple(y.shape), mask))
        if Fn.POOL_IDX_TAP is not None:
            Fn.POOL_IDX_TAP.append(pidx)
        ctx.meta = (g, B, To, Tp, kt, fused)
        ctx.bn_training = training
        if save:
            ctx.save_for_backward(gamma, y, mask, pidx, mean, invstd, *taps)
        return p

    @staticmethod
    def backward(ctx, dp):
        if not ctx.bn_training:
            raise NotImplementedError('backward through eval-mode BatchNorm is not implemented')
        gamma, y, mask, pidx, mean, invstd, *taps = ctx.saved_tensors
        g, B, To, Tp, kt, fused = ctx.meta
        Cout = y.shape[-1]
        dp = dp if dp.is_contiguous() else dp.contiguous()
        d_even = K.maxpool_bwd(dp, pidx, (B * Tp, g.Ho, g.Wo, Cout))
        da = torch.zeros((B, To, g.Ho, g.Wo, Cout), dtype=torch.float32, device=dp.device)   # odd frames: unused by pool1
        da[:, ::2] = d_even.view(B, Tp, g.Ho, g.Wo, Cout)
        dy, dgamma, dbeta = K.bn_backward(da.view(B * To, g.Ho, g.Wo, Cout), mask, y, gamma, mean, invstd, True)
        dw = None
        if ctx.needs_input_grad[1]:
            if fused:
                dw4 = K.conv_wgrad(dy, taps[0], g)                                   # (Cout, kt * kh, kw, 4)
                dw = dw4.view(Cout, kt, g.R, g.S, 4)[..., :3].permute(0, 4, 1, 2, 3)   # (Cout, 3, kt, kh, kw)
            else:
                parts = [K.conv_wgrad(dy, taps[dt], g)[..., :3] for dt in range(kt)]     # each (Cout, kh, kw, 3)
                dw = torch.stack(parts, dim=1).permute(0, 4, 1, 2, 3)                    # (Cout, 3, kt, kh, kw)
        Fn.join_side_stream(dp.device)
        return None, dw, dgamma, dbeta, None, None, None


class _PoolT2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out, sel = K.maxpool_t2_fwd(x.contiguous())
        if Fn.POOL_IDX_TAP is not None:                             # tests: which frame of each pair pool2 took
            Fn.POOL_IDX_TAP.append(sel)
        ctx.save_for_backward(sel)
        return out

    @staticmethod
    def backward(ctx, dout):
        (sel,) = ctx.saved_tensors
        return K.maxpool_t2_bwd(dout.contiguous(), sel)


@BACKBONES.register_module()
class ResNet3d(nn.Module):
    """The subset of UPSTREAM ``ResNet3d`` that configs/_base_/models/i3d_r50.py:5-15 selects.  Input (B, 3, T, H, W)."""

    arch_settings = {50: (Bottleneck3d, (3, 4, 6, 3))}

    def __init__(self, depth=50, pretrained=None, pretrained2d=True, conv1_kernel=(5, 7, 7), conv1_stride_t=2, pool1_stride_t=2,
                 conv_cfg=None, norm_eval=False, inflate=(1, 1, 1, 1), inflate_style='3x1x1', with_pool2=True,
                 zero_init_residual=False, **kwargs):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for ResNet3d on the HIP path (50 only)')
        if tuple(conv1_kernel) != (5, 7, 7) or conv1_stride_t != 2 or pool1_stride_t != 2 or inflate_style != '3x1x1' or not with_pool2:
            raise NotImplementedError('ResNet3d: only the i3d_r50 settings (conv1 5x7x7 / 2, pool1 stride_t 2, 3x1x1 inflation, pool2)')
        if conv_cfg not in (None, dict(type='Conv3d')):
            raise NotImplementedError(f'conv_cfg {conv_cfg}')
        self.depth, self.pretrained, self.pretrained2d, self.norm_eval = depth, pretrained, pretrained2d, norm_eval
        self.zero_init_residual = zero_init_residual
        block, counts = self.arch_settings[depth]
        self.conv1 = ConvModule3d(3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), act=True)
        self.conv1.conv.weight.data = self.conv1.conv.weight.data.contiguous()      # 3 input channels: repacked per tap
        self.maxpool = nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(2, 2, 2), padding=(0, 1, 1))   # holders; fused into the stem
        self.pool2 = nn.MaxPool3d(kernel_size=(2, 1, 1), stride=(2, 1, 1))
        inplanes = 64
        self.res_layers: List[str] = []
        for i, (n, planes) in enumerate(zip(counts, (64, 128, 256, 512))):
            stride = 1 if i == 0 else 2
            infl = inflate[i] if isinstance(inflate[i], (tuple, list)) else (inflate[i],) * n
            if len(infl) != n:
                raise ValueError(f'inflate[{i}] has {len(infl)} entries for {n} blocks')
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if b == 0 and (s != 1 or inplanes != planes * block.expansion):
                    down = ConvModule3d(inplanes, planes * block.expansion, (1, 1, 1), (1, s, s), act=False)
                blocks.append(block(inplanes, planes, s, bool(infl[b]), down))
                inplanes = planes * block.expansion
            name = f'layer{i + 1}'
            setattr(self, name, ResStage(*blocks))
            self.res_layers.append(name)
        self.feat_dim = inplanes

    def init_weights(self):
        if isinstance(self.pretrained, str):
            raise FileNotFoundError(f'pretrained={self.pretrained!r}: 2-D checkpoint inflation needs the torchvision file, which '
                                    f'cannot be fetched here; pass pretrained=None')
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck3d):
                    nn.init.constant_(m.conv3.bn.weight, 0.0)

    def _set_frames(self, t_layer1: int, t_rest: int):
        for li, name in enumerate(self.res_layers):
            for blk in getattr(self, name):
                for m in blk.unit_modules:
                    m.frames = t_layer1 if li == 0 else t_rest

    def forward(self, x):
        """x: (B, 3, T, H, W) -> (B, C, T', h, w) view of the NHWC frames."""
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError(f'ResNet3d: expected (B, 3, T, H, W), got {tuple(x.shape)}')
        B, _, T, H, W = x.shape
        frames = x.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)       # layout plumbing: frames in NCHW
        x4 = K.nchw3_to_nhwc4(frames.contiguous())
        stem = self.conv1
        training = stem.bn.training
        if training and stem.bn.track_running_stats:
            torch._foreach_add_([m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm3d) and m.training], 1)
        p = _Stem3dFn.apply(x4, stem.conv.weight, stem.bn.weight, stem.bn.bias, stem.bn, training, T)
        t1 = p.shape[0] // B                                             # frames per clip after conv1 and pool1
        if t1 % 2:
            raise ValueError(f'{T} input frames leave {t1} frames for pool2, which pairs them')
        self._set_frames(t1, t1 // 2)
        out = Fn.nhwc_to_nchw_view(p)
        for li, name in enumerate(self.res_layers):
            out = getattr(self, name)(out)
            if li == 0:
                out = Fn.nhwc_to_nchw_view(_PoolT2Fn.apply(Fn.nchw_view_to_nhwc(out)))
        n, c, h, w = out.shape
        return out.reshape(B, n // B, c, h, w).permute(0, 2, 1, 3, 4)    # (B, C, T', h, w) view

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm3d):
                    m.eval()
        return self


@HEADS.register_module()
class I3DHead(nn.Module):
    """UPSTREAM I3DHead: AdaptiveAvgPool3d(1) -> Dropout -> Linear; ``loss`` = UPSTREAM BaseHead.loss."""

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), spatial_type='avg', dropout_ratio=0.5,
                 init_std=0.01, multi_class=False, label_smooth_eps=0.0, topk=(1, 5), **kwargs):
        super().__init__()
        if spatial_type != 'avg' or multi_class or label_smooth_eps:
            raise NotImplementedError("I3DHead: only spatial_type='avg' without multi_class / label smoothing")
        self.num_classes, self.in_channels, self.init_std, self.topk = num_classes, in_channels, init_std, tuple(topk)
        self.loss_cls = build_loss(loss_cls)
        self.dropout = HipDropout(dropout_ratio) if dropout_ratio != 0 else None
        self.fc_cls = nn.Linear(in_channels, num_classes)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))                 # holder; the pooling runs in avgpool_fwd

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, self.init_std)
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        """x: (B, C, T, h, w) view of NHWC frames -> (B, K)."""
        B, C, T, h, w = x.shape
        frames = x.permute(0, 2, 3, 4, 1)                                # (B, T, h, w, C): the storage order
        frames = frames if frames.is_contiguous() else frames.contiguous()
        pooled = Fn.AvgPoolFn.apply(Fn.nhwc_to_nchw_view(frames.reshape(B, T * h, w, C))).reshape(B, C)
        if self.dropout is not None:
            pooled = self.dropout(pooled)
        return Fn.LinearFn.apply(pooled, self.fc_cls.weight, self.fc_cls.bias)

    def loss(self, cls_score, labels, **kwargs):
        losses = dict()
        if labels.shape == torch.Size([]):
            labels = labels.unsqueeze(0)
        if cls_score.size() != labels.size():
            acc = K.topk_acc(cls_score.detach().contiguous(), labels.contiguous())
            for i, k in enumerate(self.topk[:2]):
                losses[f'top{k}_acc'] = acc[i]
        loss_cls = self.loss_cls(cls_score, labels, **kwargs)
        if isinstance(loss_cls, dict):
            losses.update(loss_cls)
        else:
            losses['loss_cls'] = loss_cls
        return losses


@RECOGNIZERS.register_module()
class Recognizer3D(nn.Module):
    """UPSTREAM Recognizer3D: imgs (B, num_clips, 3, T, H, W); clips are folded into the batch, scores averaged over them."""

    def __init__(self, backbone, cls_head=None, neck=None, train_cfg=None, test_cfg=None):
        super().__init__()
        if neck is not None:
            raise NotImplementedError('necks are not used')
        self.backbone = build_backbone(backbone)
        self.cls_head = build_head(cls_head) if cls_head else None
        self.train_cfg = train_cfg
        self.test_cfg = dict(test_cfg) if test_cfg else {}
        self.backbone.init_weights()
        if self.cls_head is not None:
            self.cls_head.init_weights()

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    def forward_train(self, imgs, labels, **kwargs):
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        cls_score = self.cls_head(self.backbone(imgs))
        return dict(self.cls_head.loss(cls_score, labels.squeeze(), **kwargs))

    def forward_test(self, imgs):
        batches, num_segs = imgs.shape[0], imgs.shape[1]
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        cls_score = self.cls_head(self.backbone(imgs))
        mode = self.test_cfg.get('average_clips', None)
        if mode not in ['score', 'prob', None]:
            raise ValueError(f'{mode} is not supported. Currently supported ones are ["score", "prob", None]')
        if mode is None:
            return cls_score
        if mode == 'score':
            return Fn.ConsensusFn.apply(cls_score.view(batches, num_segs, -1)).squeeze(1)
        return K.softmax_mean(cls_score.contiguous(), batches, num_segs, apply_softmax=True)
