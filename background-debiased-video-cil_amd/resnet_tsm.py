"""ResNetTSM backbone behind the mmaction2 plugin surface, executed by the HIP kernels.

Module tree and ``state_dict`` keys follow UPSTREAM mmaction2 ``ResNetTSM`` exactly:
``conv1.{conv,bn}``, ``layerL.B.convK.{conv,bn}``, ``layerL.B.conv1.conv.net.weight`` (TemporalShift wrapper),
``layerL.B.downsample.{conv,bn}``.  ``nn.Conv2d`` / ``nn.BatchNorm2d`` instances are kept as *parameter
holders* so that the reference's optimizer constructor (libs/models/cil_heads/tsm.py:232-271, which raises on
unknown parameter-owning leaf modules) and checkpoints work unchanged; the arithmetic never goes through
their ``forward``.

Tensors crossing module boundaries are NCHW-shaped views of NHWC storage.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import BACKBONES


class Nhwc4Frames:
    """Output of the fused front-end: normalised frames already in the stem's NHWC4 layout."""

    def __init__(self, data: torch.Tensor, batches: int, num_segments: int):
        if data.dim() != 4 or data.shape[-1] != 4 or data.shape[0] != batches * num_segments:
            raise ValueError(f'Nhwc4Frames: bad shape {tuple(data.shape)} for B={batches}, T={num_segments}')
        self.data, self.batches, self.num_segments = data, batches, num_segments

    @property
    def shape(self):
        n, h, w, _ = self.data.shape
        return (self.batches, self.num_segments, 3, h, w)

    def size(self, dim=None):
        """Like ``Tensor.size`` of the (B, T, 3, H, W) batch this stands for (cil.py reads ``imgs.size(0)``)."""
        return self.shape if dim is None else self.shape[dim]


def _channels_last_(conv: nn.Conv2d):
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class TemporalShift(nn.Module):
    """UPSTREAM TemporalShift wrapper: keeps the ``.net`` checkpoint key.  The shift itself is fused into
    the wrapped conv's activation-tile gather (csrc/conv_mfma.hip)."""

    def __init__(self, net: nn.Conv2d, num_segments: int = 3, shift_div: int = 8):
        super().__init__()
        self.net = net
        self.num_segments = num_segments
        self.shift_div = shift_div

    @property
    def weight(self):
        return self.net.weight

    def forward(self, x):
        g = K.make_geom(x.shape[0], x.shape[2], x.shape[3], self.net.in_channels, self.net.out_channels,
                        self.net.kernel_size[0], self.net.kernel_size[1], self.net.stride[0], self.net.padding[0],
                        self.num_segments, self.net.in_channels // self.shift_div)
        return _ConvOnlyFn.apply(x, self.net.weight, g)


class _ConvOnlyFn(torch.autograd.Function):
    """Stand-alone conv (used only when a ConvModule / TemporalShift is called outside a block)."""

    @staticmethod
    def forward(ctx, x_nchw, weight, g):
        x = Fn.nchw_view_to_nhwc(x_nchw)
        ctx.save_for_backward(x, weight)
        ctx.g = g
        return Fn.nhwc_to_nchw_view(K.conv_fprop(x, Fn.weight_krsc(weight), g))

    @staticmethod
    def backward(ctx, dy_nchw):
        x, weight = ctx.saved_tensors
        dy = Fn.nchw_view_to_nhwc(dy_nchw)
        dx = Fn.nhwc_to_nchw_view(K.conv_dgrad(dy, Fn.weight_krsc(weight), ctx.g)) if ctx.needs_input_grad[0] else None
        dw = K.conv_wgrad(dy, x, ctx.g).permute(0, 3, 1, 2) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class ConvModule(nn.Module):
    """conv(bias=False) -> BN -> optional ReLU; children named ``conv`` and ``bn`` as in mmcv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, act=True):
        super().__init__()
        self.conv = _channels_last_(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False))
        self.bn = nn.BatchNorm2d(out_channels, eps=1e-5, momentum=0.1)
        self.with_activation = act

    @property
    def raw_conv(self) -> nn.Conv2d:
        return self.conv.net if isinstance(self.conv, TemporalShift) else self.conv

    @property
    def shift_div(self) -> int:
        return self.conv.shift_div if isinstance(self.conv, TemporalShift) else 0

    @property
    def num_segments(self) -> int:
        return self.conv.num_segments if isinstance(self.conv, TemporalShift) else 1

    def spec(self) -> Fn.UnitSpec:
        c = self.raw_conv
        return Fn.UnitSpec(c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], self.with_activation,
                           self.shift_div, self.num_segments)


class _ResBlock(nn.Module):
    main_names: List[str] = []

    def _finalize(self):
        self.n_main = len(self.main_names)

    @property
    def unit_modules(self) -> List[ConvModule]:
        mods = [getattr(self, n) for n in self.main_names]
        if self.downsample is not None:
            mods.append(self.downsample)
        return mods

    @property
    def unit_specs(self):
        return [m.spec() for m in self.unit_modules]

    @property
    def unit_bns(self):
        return [m.bn for m in self.unit_modules]

    def forward(self, x):
        """x: NCHW view over NHWC storage (or any NCHW tensor; converted once)."""
        xh = Fn.nchw_view_to_nhwc(x)
        params = []
        for m in self.unit_modules:
            params += [m.raw_conv.weight, m.bn.weight, m.bn.bias]
        training = self.unit_bns[0].training
        out = Fn.ResBlockFn.apply(xh, self, training, *params)
        return Fn.nhwc_to_nchw_view(out)


class ResStage(nn.Sequential):
    """UPSTREAM ``ResNet.layerN``: the blocks of one stage.  Same module tree and ``state_dict`` keys as an
    ``nn.Sequential``; the forward runs the stage as one autograd node (``ResStageFn``) so that kernels can fuse across
    block boundaries.  If anything hooks an individual block (forward / pre-forward / backward hooks) the stage falls back
    to block-by-block execution, where every block output is a regular autograd tensor."""

    def _blocks_are_plain(self):
        for b in self:
            if not isinstance(b, _ResBlock):
                return False
            if b._forward_hooks or b._forward_pre_hooks or b._backward_hooks or getattr(b, '_backward_pre_hooks', None):
                return False
        return True

    def forward(self, x):
        # links to the neighbouring stages: set by the backbone for this call only, taken (and cleared) before anything can fail or
        # fall back, so that a stage never keeps the producer's saved tensors alive (nor hands them to a deepcopy of the model)
        in_link, out_link = getattr(self, '_in_link', None), getattr(self, '_out_link', None)
        self._in_link = self._out_link = None
        if not Fn.FUSE_STAGE or len(self) < 2 or not self._blocks_are_plain():
            return super().forward(x)
        xh = Fn.nchw_view_to_nhwc(x)
        blocks = list(self)
        params = []
        for b in blocks:
            for m in b.unit_modules:
                params += [m.raw_conv.weight, m.bn.weight, m.bn.bias]
        flags = {b.unit_bns[0].training for b in blocks}
        if len(flags) != 1:
            return super().forward(x)
        out = Fn.ResStageFn.apply(xh, blocks, flags.pop(), in_link, out_link, *params)
        return Fn.nhwc_to_nchw_view(out)


class BasicBlock(_ResBlock):
    expansion = 1
    main_names = ['conv1', 'conv2']

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvModule(inplanes, planes, 3, stride, 1, act=True)
        self.conv2 = ConvModule(planes, planes, 3, 1, 1, act=False)
        self.downsample = downsample
        self._finalize()


class Bottleneck(_ResBlock):
    expansion = 4
    main_names = ['conv1', 'conv2', 'conv3']

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvModule(inplanes, planes, 1, 1, 0, act=True)
        self.conv2 = ConvModule(planes, planes, 3, stride, 1, act=True)     # style='pytorch'
        self.conv3 = ConvModule(planes, planes * 4, 1, 1, 0, act=False)
        self.downsample = downsample
        self._finalize()


def _has_hooks(m: nn.Module, pre: bool) -> bool:
    """pre=False: anything that observes the module's OUTPUT or its gradient; pre=True: anything that observes its INPUT."""
    if pre:
        return bool(m._forward_pre_hooks or getattr(m, '_backward_pre_hooks', None))
    return bool(m._forward_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None))


def _global_module_hooks() -> bool:
    mod = torch.nn.modules.module
    return any(bool(getattr(mod, n, None)) for n in ('_global_forward_hooks', '_global_forward_pre_hooks', '_global_backward_hooks',
                                                      '_global_backward_pre_hooks'))


@BACKBONES.register_module()
class ResNetTSM(nn.Module):
    """UPSTREAM mmaction2 ResNetTSM(depth, num_segments=8, is_shift=True, shift_div=8,
    shift_place='blockres'); config site: configs/ucf101/bgmix_plus_randAug/...:59-65."""

    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3))}

    def __init__(self, depth, num_segments=8, is_shift=True, non_local=(0, 0, 0, 0), non_local_cfg=None, shift_div=8,
                 shift_place='blockres', temporal_pool=False, pretrained=None, norm_eval=False, **kwargs):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if shift_place != 'blockres' or temporal_pool or any(non_local):
            raise NotImplementedError('only shift_place="blockres" without temporal_pool / non_local is on the hot path '
                                      '(all 84 CIL configs; SURVEY.md section 8)')
        self.depth, self.num_segments, self.is_shift, self.shift_div = depth, num_segments, is_shift, shift_div
        self.pretrained, self.norm_eval = pretrained, norm_eval
        block, counts = self.arch_settings[depth]
        self.conv1 = ConvModule(3, 64, 7, 2, 3, act=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)       # holder only; fused into StemFn
        inplanes = 64
        self.res_layers = []
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
            name = f'layer{i + 1}'
            setattr(self, name, ResStage(*blocks))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._shift_made = False

    # ---- initialisation (UPSTREAM ResNet.init_weights + ResNetTSM.make_temporal_shift) ----
    def init_weights(self):
        if isinstance(self.pretrained, str):
            self._load_torchvision_checkpoint(self.pretrained)
        elif self.pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1.0)
                    nn.init.constant_(m.bias, 0.0)
        else:
            raise TypeError('pretrained must be a str or None')
        if self.is_shift and not self._shift_made:
            self.make_temporal_shift()

    def make_temporal_shift(self):
        for name in self.res_layers:
            for blk in getattr(self, name):
                blk.conv1.conv = TemporalShift(blk.conv1.conv, num_segments=self.num_segments, shift_div=self.shift_div)
        self._shift_made = True

    def _load_torchvision_checkpoint(self, path: str):
        """Local torchvision-format file only (no network here).  Key mapping per SURVEY Appendix A:
        ``conv1.weight -> conv1.conv.weight``, ``bn1.* -> conv1.bn.*``, ``layerL.B.convK.weight ->
        layerL.B.convK.conv.weight``, ``bnK -> convK.bn``, ``downsample.0 -> downsample.conv``, ``.1 -> .bn``."""
        if path.startswith(('http://', 'https://', 'torchvision://')):
            raise FileNotFoundError(f'pretrained={path!r}: remote checkpoints cannot be fetched here; pass a local file or None')
        sd = torch.load(path, map_location='cpu', weights_only=True)
        sd = sd.get('state_dict', sd)
        mapped = {}
        for k, v in sd.items():
            if k.startswith('fc.'):
                continue
            parts = k.split('.')
            if parts[0] == 'conv1':
                nk = 'conv1.conv.' + parts[1]
            elif parts[0] == 'bn1':
                nk = 'conv1.bn.' + parts[1]
            elif parts[0].startswith('layer'):
                l, b, m = parts[0], parts[1], parts[2]
                if m.startswith('conv'):
                    nk = f'{l}.{b}.{m}.conv.{parts[3]}'
                elif m.startswith('bn'):
                    nk = f'{l}.{b}.conv{m[2:]}.bn.{parts[3]}'
                elif m == 'downsample':
                    nk = f'{l}.{b}.downsample.{"conv" if parts[3] == "0" else "bn"}.{parts[4]}'
                else:
                    continue
            else:
                continue
            mapped[nk] = v
        missing = self.load_state_dict(mapped, strict=False)
        if missing.unexpected_keys:
            raise KeyError(f'unexpected keys in checkpoint: {missing.unexpected_keys[:5]}')

    # ---- forward ------------------------------------------------------------------------
    def _bn_modules(self):
        return [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]

    def forward(self, x):
        """x: (N,3,H,W) fp32 NCHW (the reference's boundary) or ``Nhwc4Frames`` from the fused front-end.
        Returns the layer4 feature map as an (N,C,h,w) view of NHWC storage."""
        if isinstance(x, Nhwc4Frames):
            x4 = x.data
        else:
            x4 = K.nchw3_to_nhwc4(x.contiguous())
        stem = self.conv1
        training = stem.bn.training
        if training and stem.bn.track_running_stats:
            torch._foreach_add_([b.num_batches_tracked for b in self._bn_modules() if b.training], 1)
        p = Fn.StemFn.apply(x4, stem.conv.weight, stem.bn.weight, stem.bn.bias, stem.bn, training)
        out = Fn.nhwc_to_nchw_view(p)
        link = None
        stages = [getattr(self, name) for name in self.res_layers]
        for i, stage in enumerate(stages):
            if isinstance(stage, ResStage):
                stage._in_link = link
                # this stage's output is private to the next stage unless something can see it on the way: a forward / backward hook
                # on the stage module (OutputHook for the feature-KD terms), a pre-hook of the next stage (which receives the same
                # tensor), or a global module hook.  Only a private output lets the next stage's backward take BatchNorm statistics
                # for this one.
                nxt = stages[i + 1] if i + 1 < len(stages) else None
                private = (nxt is not None and isinstance(nxt, ResStage) and not _has_hooks(stage, pre=False) and not _has_hooks(nxt, pre=True)
                           and not _global_module_hooks())
                link = Fn.StageLink() if (training and private) else None
                stage._out_link = link
            else:
                link = None
            try:
                out = stage(out)
            finally:
                if isinstance(stage, ResStage):
                    stage._in_link = stage._out_link = None      # (a stage that raised or was replaced never took them)
        return out

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self
