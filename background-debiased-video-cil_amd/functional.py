"""autograd glue: every differentiable step of the hot path is a ``torch.autograd.Function`` whose
forward/backward only enqueue kernels of ``libbdvcil_hip.so`` (through ``kernels``).

Granularity is chosen so that no gradient junction inside a residual block is left to autograd
(the identity-path add and its ReLU mask are fused into the conv1 dgrad epilogue):

* ``StemFn``      NHWC4 frames -> conv7x7/2 + BN + ReLU + maxpool3x3/2
* ``ResBlockFn``  one BasicBlock / Bottleneck incl. temporal shift, BN, ReLU, residual
* ``AvgPoolFn``, ``DropoutFn``, ``LSCFn``, ``LinearFn``, ``ConsensusFn``
* ``LSCLossFn``, ``SoftCEFn``, ``KDMSEFn``

Internal activation layout is NHWC fp32; module boundaries expose NCHW *views* of the same
storage (``nhwc_to_nchw_view``), so hooks and feature-distillation see mmaction-shaped tensors.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import kernels as K


# ---------------------------------------------------------------------------------------------
# side stream for weight gradients
# ---------------------------------------------------------------------------------------------
# In backward, wgrad(l) only feeds the optimizer while BatchNorm-backward(l) -> dgrad(l) -> BatchNorm-backward(l-1) -> ... is the
# critical chain.  The wgrads are launched on a second HIP stream: the HBM-bound BatchNorm-backward passes of the chain then
# run beside the MFMA-bound wgrad of the layer above (the 8-wave conv kernels occupy one workgroup per CU and leave wave
# slots free), and the CUs a dgrad's last partial round of workgroups leaves idle are filled by wgrad workgroups.
# ``join_side_stream`` must run before gradients are consumed (FusedSGD.step / clip / GradAllReducer / StemFn.backward do it).
# Measured on one box, alternating runs (profiles/r02_side_stream.txt): 535.1 / 535.0 clips/s without, 551.5 / 550.5 with.
# BDVCIL_WGRAD_SIDE_STREAM=0 puts everything back on one stream (one batched split-K reduction per stage then).
import collections as _collections
import os as _os
_SIDE = {'enabled': _os.environ.get('BDVCIL_WGRAD_SIDE_STREAM', '1') != '0', 'streams': {}, 'pending': {}, 'held': {}}
# Lifetime of the operands of a weight gradient running on the side stream.  torch's record_stream() defers the release of a
# block until the side stream's work at release time has finished; the host enqueues a whole step ahead of the device, so none of
# those blocks was ever reusable inside a step and the caching allocator went back to hipMalloc for every gradient and
# activation tensor: 115 GB reserved for 24 GB allocated at batch 32, 225 GB for 46 GB at batch 64 -- where steps then ran
# 1.3 - 2.5 x slower in some processes (gpurun_out/lag*.log: 94 ms against 70 ms per step).  Instead the operands stay referenced
# in a FIFO until the main stream has waited for the side stream (join_side_stream: the end of the backward pass, or a gradient
# bucket becoming ready), after which they return to the pool in plain stream order: 46 GB reserved at batch 32, no extra
# synchronisation, step time level or better (profiles/r02_ab_streams.txt).  BDVCIL_SIDE_LAG bounds the FIFO: the main stream
# waits for the weight gradient that many launches back before dropping its operands (64 = never inside a ResNet-50 step;
# smaller values trade 1 % of step time for a few GB); 0 = record_stream().
SIDE_LAG = int(_os.environ.get('BDVCIL_SIDE_LAG', '64'))


# BatchNorm-backward statistics of a unit taken in the epilogue of the dgrad that produces its output gradient
# (bdv_conv_dgrad with a bdv_bn_stat_fuse): removes the separate pass over dout, y and the mask for the inner units.
FUSE_BN_STATS = _os.environ.get('BDVCIL_FUSE_BN_STATS', '1') != '0'
# A unit's BatchNorm + ReLU applied in the loaders of the conv that consumes it (fprop and weight gradient) instead of in an apply
# pass of its own: conv1 -> conv2 and conv2 -> conv3 inside a block (the block output is read by two consumers and stays a pass).
# The activation and its ReLU mask are then never written; the backward kernels derive the sign from the conv output.
PRE_BN = _os.environ.get('BDVCIL_PRE_BN', '0') == '1'      # off: measured 0.6 ms per step SLOWER; saves ~2.8 GB
# The forward half of it alone: the consumer conv applies the producer's BatchNorm + ReLU in its loader and starts right away, while
# the apply pass that writes the activation and the mask for the BACKWARD pass runs beside it on the side stream (an HBM-bound pass
# beside an MFMA-bound conv).  The backward pass is the default one.  BDVCIL_PRE_BN=fwd.
PRE_BN_FWD = _os.environ.get('BDVCIL_PRE_BN', '0') == 'fwd'
SAVED_PER_UNIT = 7      # y, activation | None, mean, invstd, mask | None, scale | None, shift | None
# A whole stage as one autograd node (ResStageFn): lets the statistics fusion above reach the block outputs.
FUSE_STAGE = _os.environ.get('BDVCIL_FUSE_STAGE', '1') != '0'
# one split-K reduction launch per autograd node (stage / block) instead of one per conv
BATCH_WGRAD_REDUCE = _os.environ.get('BDVCIL_BATCH_WGRAD_REDUCE', '1') != '0'


# Test hook: when set to a list, every training-mode forward appends the 1-bit ReLU masks it writes, in execution order
# (stem, then per block: unit 0, unit 1, ..., block output); the parity tests compare them with the CPU reference's signs.
RELU_MASK_TAP = None
POOL_IDX_TAP = None      # tests: the stem max-pool's arg-max codes (3 r + s per output element), same purpose


def set_side_stream_enabled(flag: bool):
    _SIDE['enabled'] = bool(flag)


def side_stream_enabled() -> bool:
    return _SIDE['enabled']


def side_stream(device) -> torch.cuda.Stream:
    return _side_stream(device)[1]


def _side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE['streams'].get(idx)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE['streams'][idx] = st
    return idx, st


def join_side_stream(device=None):
    """Make the current stream wait for every wgrad launched on the side stream so far."""
    for idx, ev in list(_SIDE['pending'].items()):
        if device is None or (device.index if device.index is not None else torch.cuda.current_device()) == idx:
            torch.cuda.current_stream(idx).wait_event(ev)
            del _SIDE['pending'][idx]
            held = _SIDE['held'].get(idx)
            if held:
                held.clear()        # the event is the side stream's latest: the current stream is now behind all of its work


class wgrad_batch:
    """Inside ``with wgrad_batch():`` the split-K reductions of all weight gradients are deferred and run as one launch
    when the block ends (a stage's backward: one reduce kernel instead of one per conv).  The returned dw tensors are
    filled at that point, i.e. before the autograd node hands them out."""
    current = None

    def __enter__(self):
        self.items, self.outer = [], wgrad_batch.current
        if BATCH_WGRAD_REDUCE:
            wgrad_batch.current = self
        return self

    def __exit__(self, exc_type, exc, tb):
        wgrad_batch.current = self.outer
        if exc_type is None and self.items:
            if _SIDE['enabled'] and self.items[0][0].is_cuda:     # the partial products were launched on the side stream
                idx, side = _side_stream(self.items[0][0].device)
                with torch.cuda.stream(side):
                    K.wgrad_reduce_batched(self.items)
                    done = torch.cuda.Event()
                    done.record(side)
                _SIDE['pending'][idx] = done
            else:
                K.wgrad_reduce_batched(self.items)
        self.items = []
        return False


# How the weight gradient of a conv gets its activation operand when the producer's apply pass never ran (PRE_BN):
# 'recompute' (default): one bn_apply on the stream the weight gradient runs on (the side stream: an HBM-bound pass beside the
# main chain's MFMA-bound dgrads), then the plain kernel; 'loader': the producer's BatchNorm + ReLU in the weight-gradient kernel's
# own loader (measured slower in the step: that loader already splits both operands).
PRE_BN_WGRAD = _os.environ.get('BDVCIL_PRE_BN_WGRAD', 'recompute')


def wgrad_overlapped(dy: torch.Tensor, inp: torch.Tensor, geom, pre_bn=None) -> torch.Tensor:
    """Weight gradient of one conv: deferred reduction inside a ``wgrad_batch``, otherwise K.conv_wgrad (optionally on the
    side stream)."""
    batch = wgrad_batch.current
    recompute = pre_bn is not None and PRE_BN_WGRAD != 'loader'
    if not _SIDE['enabled']:
        if recompute:
            inp, pre_bn = K.bn_apply(inp, pre_bn[0], pre_bn[1], None, True), None
        if batch is not None:
            slab, dw = K.conv_wgrad_partial(dy, inp, geom, pre_bn=pre_bn)
            batch.items.append((slab, dw))
            return dw
        return K.conv_wgrad(dy, inp, geom, pre_bn=pre_bn)
    main = torch.cuda.current_stream(dy.device)
    idx, side = _side_stream(dy.device)
    dw = torch.empty((geom.Cout, geom.R, geom.S, geom.Cin), dtype=torch.float32, device=dy.device)   # owned by main
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    raw = inp                       # what the side stream READS (allocated on main): held / registered below, also when `inp` is rebound
    with torch.cuda.stream(side):
        if recompute:               # the activation exists only for the duration of this weight gradient
            inp = K.bn_apply(inp, pre_bn[0], pre_bn[1], None, True)
            pre_bn = None
        if batch is not None:       # split-K slabs now, one reduction launch (on the side stream) when the stage ends
            slab, _ = K.conv_wgrad_partial(dy, inp, geom, dw=dw, pre_bn=pre_bn)
            batch.items.append((slab, dw))
        else:
            K.conv_wgrad(dy, inp, geom, dw=dw, beta=0.0, ws_tag='wgrad_side', pre_bn=pre_bn)
        done = torch.cuda.Event()
        done.record(side)
    if SIDE_LAG > 0:
        held = _SIDE['held'].setdefault(idx, _collections.deque())
        held.append((done, (dy, raw, inp)))
        while len(held) > SIDE_LAG:
            ev, _ = held.popleft()
            main.wait_event(ev)     # (long finished: the side stream runs beside the main chain, not behind it)
    else:
        for t in (dy, raw, inp, dw):
            t.record_stream(side)
    _SIDE['pending'][idx] = done
    return dw


# The downsample branch of a block (1x1 conv + its BatchNorm statistics) depends only on the block input: it can run on the
# side stream beside conv1 / conv2 / conv3 of the main branch and is joined before the block-output kernel.
DS_SIDE = _os.environ.get('BDVCIL_DS_SIDE', '1') != '0'


def run_on_side_stream(fn, device):
    """Run ``fn()`` (kernel launches only) on the side stream after everything enqueued on the current stream so far;
    returns (result, event).  The caller makes the consumer stream wait for the event; tensors created inside are
    registered with the current stream as well, so their memory is not recycled under the consumer."""
    main = torch.cuda.current_stream(device)
    _, side = _side_stream(device)
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        K.WS_TAG_SUFFIX = '_side'
        try:
            out = fn()
        finally:
            K.WS_TAG_SUFFIX = ''
        done = torch.cuda.Event()
        done.record(side)
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        if torch.is_tensor(t):
            t.record_stream(main)
    return out, done


def nhwc_to_nchw_view(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def nchw_view_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def weight_krsc(w: torch.Tensor) -> torch.Tensor:
    """OIHW parameter (channels_last storage) -> (Cout,R,S,Cin) contiguous view (copy only if the
    parameter is not channels_last, e.g. a foreign checkpoint tensor assigned by hand).  A 5-D Conv3d weight
    (Cout, Cin, kt, kh, kw) in channels_last_3d storage gives (Cout, kt, 1, Cin) for a kt x 1 x 1 filter and
    (Cout, kh, kw, Cin) for a 1 x kh x kw filter (the two kinds an I3D bottleneck has)."""
    if w.dim() == 5:
        co, ci, kt, kh, kw = w.shape
        if kt > 1 and (kh > 1 or kw > 1):
            raise NotImplementedError('weight_krsc: a filter that is both temporal and spatial (only the I3D stem has one)')
        v = w.permute(0, 2, 3, 4, 1)
        v = v if v.is_contiguous() else v.contiguous()
        return v.reshape(co, kt, 1, ci) if kt > 1 else v.reshape(co, kh, kw, ci)
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def grad_like_weight(dw_krsc: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """(Cout,R,S,Cin) kernel output -> a gradient of the parameter's logical shape (a view: its storage is the parameter's)."""
    if w.dim() == 5:
        co, ci, kt, kh, kw = w.shape
        return dw_krsc.reshape(co, kt, kh, kw, ci).permute(0, 4, 1, 2, 3)
    return dw_krsc.permute(0, 3, 1, 2)


class UnitSpec:
    """Static description of one conv+BN(+ReLU) site inside a block."""

    def __init__(self, cin, cout, k, stride, pad, relu, shift_div=0, num_segments=1):
        self.cin, self.cout, self.k, self.stride, self.pad, self.relu = cin, cout, k, stride, pad, relu
        self.fold = cin // shift_div if shift_div else 0
        self.T = num_segments if shift_div else 1

    def geom(self, N, H, W):
        return K.make_geom(N, H, W, self.cin, self.cout, self.k, self.k, self.stride, self.pad, self.T, self.fold)

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1


class TemporalUnitSpec(UnitSpec):
    """kt x 1 x 1 convolution + BN(+ReLU) of an I3D bottleneck: per frame nothing changes spatially; the conv runs on the
    [B][T][H*W][C] view of the frames (``frames`` per clip = T)."""

    def __init__(self, cin, cout, kt, relu, frames):
        super().__init__(cin, cout, 1, 1, 0, relu)
        self.kt, self.frames = kt, frames

    def geom(self, N, H, W):
        if N % self.frames:
            raise ValueError(f'{N} frames are not whole clips of {self.frames}')
        return K.make_temporal_geom(N // self.frames, self.frames, H, W, self.cin, self.cout, self.kt)

    def out_hw(self, H, W):
        return H, W


def _conv_bn_forward(x, w_krsc, g, bn, gamma, beta, training, pre_bn=None):
    """conv + BatchNorm statistics -> (y, mean, invstd, scale, shift) in training mode: the batch statistics come out
    of the conv epilogue (no extra pass over y) and running stats are updated in place.  ``pre_bn``: (scale, shift) of the
    producing unit when x is its raw conv output (its BatchNorm + ReLU is applied in this conv's loader)."""
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm momentum=None (cumulative average) is not supported by the HIP path')
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    y, part = K.conv_fprop(x, w_krsc, g, bn_stats=True, pre_bn=pre_bn)
    mean, invstd, scale, shift = K.bn_train_finalize(part, g.N * g.Ho * g.Wo, gamma, beta, bn.eps, bn.momentum, rm, rv)
    return y, mean, invstd, scale, shift


def _conv_bn_eval(x, w_krsc, g, bn, gamma, beta, res, relu):
    """Eval-mode conv + BatchNorm (+ residual) (+ ReLU) in one kernel: the running statistics fold into a per-channel
    scale / shift applied in the conv epilogue."""
    scale, shift = K.bn_eval_params(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
    return K.conv_fprop(x, w_krsc, g, affine=(scale, shift, res, relu))


def _bn_wgrad_backward(dout, mask, y, gamma, mean, invstd, inp, geom, need_dw, stat_partial=None, relu_affine=None, pre_bn=None):
    """BatchNorm(+ReLU) backward of one conv+BN unit followed by the conv's wgrad -> (dy, dgamma, dbeta, dw | None).
    ``relu_affine``: (scale, shift) of THIS unit when its mask was never written (the sign is derived from y);
    ``pre_bn``: (scale, shift) of the unit that produced ``inp`` when ``inp`` is that unit's raw conv output."""
    dy, dg, db = K.bn_backward(dout, mask, y, gamma, mean, invstd, True, stat_partial=stat_partial, relu_affine=relu_affine)
    dw = wgrad_overlapped(dy, inp, geom, pre_bn=pre_bn) if need_dw else None
    return dy, dg, db, dw


class StemFn(torch.autograd.Function):
    """UPSTREAM ResNet.conv1 (7x7/2 ConvModule) + ResNet.maxpool on NHWC4 input."""

    @staticmethod
    def forward(ctx, x4, weight, gamma, beta, bn, training):
        N, H, W, _ = x4.shape
        g = K.make_geom(N, H, W, 4, weight.shape[0], weight.shape[2], weight.shape[3], 2, weight.shape[2] // 2)
        w4 = torch.zeros((weight.shape[0], weight.shape[2], weight.shape[3], 4), dtype=torch.float32, device=x4.device)
        w4[..., :3] = weight.detach().permute(0, 2, 3, 1)           # 37 KB repack, plumbing
        save = training and any(ctx.needs_input_grad)
        if training:
            y, mean, invstd, scale, shift = _conv_bn_forward(x4, w4, g, bn, gamma, beta, training)
            # BN apply + ReLU + max-pool + ReLU sign mask in one pass; the activation itself is never materialised
            p, idx, mask = K.bn_relu_maxpool_fwd(y, scale, shift, out_dtype=K.ACT_DTYPE)
            if RELU_MASK_TAP is not None:
                RELU_MASK_TAP.append((tuple(y.shape), mask))
            if POOL_IDX_TAP is not None:
                POOL_IDX_TAP.append(idx)
            a_shape = tuple(y.shape)
        else:
            a = _conv_bn_eval(x4, w4, g, bn, gamma, beta, None, True)
            p, idx = K.maxpool_fwd(a, out_dtype=K.ACT_DTYPE)
            a_shape = tuple(a.shape)
            y = mask = mean = invstd = None
        ctx.g = g
        ctx.bn_training = training
        ctx.a_shape = a_shape
        if save:
            ctx.save_for_backward(x4, gamma, y, mask, idx, mean, invstd)
        return p

    @staticmethod
    def backward(ctx, dp):
        if not ctx.bn_training:
            raise NotImplementedError('backward through eval-mode BatchNorm is not implemented (norm_eval=False in all CIL configs)')
        x4, gamma, y, mask, idx, mean, invstd = ctx.saved_tensors
        dp = dp if dp.is_contiguous() else dp.contiguous()
        # max-pool backward + BN/ReLU backward in one go: the 822 MB gradient of the stem activation is never written
        dy, dgamma, dbeta = K.bn_backward_maxpool(dp, idx, mask, y, gamma, mean, invstd)
        dw = None
        if ctx.needs_input_grad[1]:
            dw4 = K.conv_wgrad(dy, x4, ctx.g)
            dw = dw4[..., :3].permute(0, 3, 1, 2)
        join_side_stream(dp.device)          # the stem is the last backward node: all wgrads are visible after it
        return None, dw, dgamma, dbeta, None, None


def _block_forward(x, blk, training, params, save):
    """Forward of one residual block (UPSTREAM BasicBlock / Bottleneck, shift_place='blockres') on NHWC storage.
    Returns (out, saved tensors for backward, geoms)."""
    units: List[UnitSpec] = blk.unit_specs
    bns = blk.unit_bns
    n_main = blk.n_main
    has_down = len(units) > n_main
    N, H, W, _ = x.shape
    saved = [x]
    geoms = []
    # identity path
    if has_down:
        u = units[n_main]
        wd, gd, bd = params[3 * n_main:3 * n_main + 3]
        g = u.geom(N, H, W)
        ds_event = None
        if training:
            # the downsample BatchNorm is applied inside the block-output kernel (res_affine): no identity tensor
            if DS_SIDE and _SIDE['enabled'] and x.is_cuda:
                (yd, mean_d, invstd_d, sc_d, sh_d), ds_event = run_on_side_stream(
                    lambda: _conv_bn_forward(x, weight_krsc(wd), g, bns[n_main], gd, bd, training), x.device)
                x.record_stream(_side_stream(x.device)[1])
            else:
                yd, mean_d, invstd_d, sc_d, sh_d = _conv_bn_forward(x, weight_krsc(wd), g, bns[n_main], gd, bd, training)
            identity, id_affine = yd, (sc_d, sh_d)
        else:
            identity, id_affine = _conv_bn_eval(x, weight_krsc(wd), g, bns[n_main], gd, bd, None, False), None
    else:
        identity, id_affine = x, None
    cur = x
    side_events = []        # apply passes running on the side stream (PRE_BN_FWD): joined before the block returns
    pending = None          # (scale, shift) of the previous unit when `cur` is its raw conv output
    h, w_ = H, W
    for i in range(n_main):
        u = units[i]
        wt, gm, bt = params[3 * i:3 * i + 3]
        g = u.geom(N, h, w_)
        geoms.append(g)
        last = i == n_main - 1
        if not training:     # eval: conv + folded BatchNorm (+ identity) + ReLU in one kernel
            h, w_ = u.out_hw(h, w_)
            cur = _conv_bn_eval(cur, weight_krsc(wt), g, bns[i], gm, bt, identity if last else None, True).view(N, h, w_, u.cout)
            continue
        y, mean, invstd, sc, sh = _conv_bn_forward(cur, weight_krsc(wt), g, bns[i], gm, bt, training, pre_bn=pending)
        y = y.view(N, *u.out_hw(h, w_), u.cout)          # frames view (a temporal geometry describes another view of it)
        if last and has_down and training and ds_event is not None:
            torch.cuda.current_stream(x.device).wait_event(ds_event)      # the identity branch is needed from here on
        h2, w2 = u.out_hw(h, w_)
        # the next unit of the main branch can apply this unit's BatchNorm + ReLU in its own loaders: no apply pass here
        can_pre = not last and (PRE_BN or PRE_BN_FWD) and K.fprop_pre_ok(units[i + 1].geom(N, h2, w2))
        defer = can_pre and PRE_BN
        if can_pre and not defer and save and _SIDE['enabled'] and y.is_cuda:
            # forward-only form: the next conv reads the raw output; activation + mask for the backward pass come from the side stream
            (a, mask), ev = run_on_side_stream(lambda: K.bn_apply(y, sc, sh, None, True, want_mask=True), y.device)
            side = _side_stream(y.device)[1]
            for t in (y, sc, sh):
                t.record_stream(side)
            main = torch.cuda.current_stream(y.device)
            a.record_stream(main)
            mask.record_stream(main)
            side_events.append(ev)
            saved += [y, a, mean, invstd, mask, None, None]
            if RELU_MASK_TAP is not None:
                RELU_MASK_TAP.append((tuple(y.shape), mask))
            cur, pending = y, (sc, sh)
        elif defer:
            if RELU_MASK_TAP is not None:       # tests read every ReLU's sign bits: produce them on the side
                RELU_MASK_TAP.append((tuple(y.shape), K.bn_apply(y, sc, sh, None, True, want_mask=True)[1]))
            if save:
                saved += [y, None, mean, invstd, None, sc, sh]
            cur, pending = y, (sc, sh)
        else:
            if save:
                a, mask = K.bn_apply(y, sc, sh, identity if last else None, True, want_mask=True,
                                     res_affine=id_affine if last else None)
                saved += [y, a, mean, invstd, mask, None, None]
                if RELU_MASK_TAP is not None:
                    RELU_MASK_TAP.append((tuple(y.shape), mask))
            else:
                a = K.bn_apply(y, sc, sh, identity if last else None, True, res_affine=id_affine if last else None)
            cur, pending = a, None
        h, w_ = h2, w2
    for ev in side_events:  # (long finished by now: an apply pass is a tenth of the conv that ran beside it)
        torch.cuda.current_stream(x.device).wait_event(ev)
    if save and has_down:
        saved += [yd, mean_d, invstd_d]
        geoms.append(units[n_main].geom(N, H, W))
    return cur, (saved if save else []), geoms


def _block_out_stats(saved, n_main):
    """(y, mask, mean, invstd) of a block's last main unit: what a dgrad epilogue needs to take the BatchNorm-backward
    statistics of the gradient it writes into that block's output."""
    k = n_main - 1
    q = SAVED_PER_UNIT * k
    return saved[1 + q], saved[5 + q], saved[3 + q], saved[4 + q]


def _block_backward(saved, params, geoms, n_main, has_down, dout, need_params, need_dx, out_stat_partial=None,
                    prev_stats=None):
    """Backward of one residual block.  ``out_stat_partial``: tile sums of the BatchNorm-backward statistics of ``dout``
    against this block's last unit, when the producer of ``dout`` already took them.  ``prev_stats``: statistics
    operands of the PREVIOUS block's last unit; the dgrad that writes dx then reduces them in its epilogue.
    Returns (dx, parameter gradients, partial for the previous block | None)."""
    x = saved[0]
    Q = SAVED_PER_UNIT
    ys = [saved[1 + Q * i] for i in range(n_main)]
    acts = [saved[2 + Q * i] for i in range(n_main)]
    means = [saved[3 + Q * i] for i in range(n_main)]
    invstds = [saved[4 + Q * i] for i in range(n_main)]
    masks = [saved[5 + Q * i] for i in range(n_main)]
    # units whose apply pass never ran (PRE_BN): (scale, shift) instead of an activation and a mask
    affs = [(saved[6 + Q * i], saved[7 + Q * i]) if saved[6 + Q * i] is not None else None for i in range(n_main)]
    out_mask = masks[-1]
    dout = dout if dout.is_contiguous() else dout.contiguous()
    grads: List[Optional[torch.Tensor]] = [None] * len(params)

    # main branch, last unit first.  ``d`` is the gradient w.r.t. the unit's (post-ReLU) output.
    d, part = dout, out_stat_partial
    for i in range(n_main - 1, -1, -1):
        wt, gm = params[3 * i], params[3 * i + 1]
        # this conv's input: the previous unit's activation, or its raw conv output + (scale, shift) for the loader
        inp = (acts[i - 1] if affs[i - 1] is None else ys[i - 1]) if i > 0 else x
        dy, dg, db, dw = _bn_wgrad_backward(d, masks[i], ys[i], gm, means[i], invstds[i], inp, geoms[i], need_params[3 * i],
                                            stat_partial=part, relu_affine=affs[i], pre_bn=affs[i - 1] if i > 0 else None)
        grads[3 * i + 1], grads[3 * i + 2] = dg, db
        if dw is not None:
            grads[3 * i] = grad_like_weight(dw, wt)
        part = None
        if i > 0:
            gi = geoms[i]
            if FUSE_BN_STATS and (gi.stride == 1 or (gi.R > 1 and gi.S > 1 and gi.fold == 0)):
                # this dgrad produces the gradient entering unit i-1's BN+ReLU: take its statistics in the epilogue
                # (stride 2: a 3x3 filter reaches every input pixel, one block of partial rows per parity class)
                prev = (ys[i - 1], masks[i - 1], means[i - 1], invstds[i - 1])
                d, part = K.conv_dgrad(dy, weight_krsc(wt), gi, bn_stats=prev if affs[i - 1] is None else prev + (affs[i - 1],))
            else:
                d = K.conv_dgrad(dy, weight_krsc(wt), gi)
            d = d.view_as(inp)                     # frames view (the geometry of a temporal conv names another view)
        else:
            dy_first = dy

    dx, prev_partial = None, None
    stats = prev_stats if (FUSE_BN_STATS and prev_stats is not None and geoms[0].stride == 1) else None
    if has_down:
        yd, mean_d, invstd_d = saved[1 + SAVED_PER_UNIT * n_main:4 + SAVED_PER_UNIT * n_main]
        wd, gd = params[3 * n_main], params[3 * n_main + 1]
        gdn = geoms[n_main]
        # gradient entering the downsample BN is dout * (out > 0): same mask as the block output
        dyd, dgd, dbd, dwd = _bn_wgrad_backward(dout, out_mask, yd, gd, mean_d, invstd_d, x, gdn, need_params[3 * n_main])
        grads[3 * n_main + 1], grads[3 * n_main + 2] = dgd, dbd
        if dwd is not None:
            grads[3 * n_main] = grad_like_weight(dwd, wd)
        if need_dx:
            dx_id = K.conv_dgrad(dyd, weight_krsc(wd), gdn)
            dx = K.conv_dgrad(dy_first, weight_krsc(params[0]), geoms[0], add_src=dx_id, bn_stats=stats)
    elif need_dx:
        # identity path: dout * (out > 0), fused into the conv1 dgrad epilogue
        dx = K.conv_dgrad(dy_first, weight_krsc(params[0]), geoms[0], add_src=dout, add_mask_src=out_mask, bn_stats=stats)
    if stats is not None and dx is not None:
        dx, prev_partial = dx
    if dx is not None:
        dx = dx.view_as(x)
    return dx, grads, prev_partial


_EVAL_BN_BACKWARD = 'backward through eval-mode BatchNorm is not implemented (norm_eval=False in all CIL configs)'


class ResBlockFn(torch.autograd.Function):
    """One residual block as an autograd node (used when a stage cannot run as one node, e.g. hooks on its blocks)."""

    @staticmethod
    def forward(ctx, x, blk, training, *params):
        save = training and any(ctx.needs_input_grad)
        out, saved, geoms = _block_forward(x, blk, training, params, save)
        if save:
            ctx.save_for_backward(*saved, *params)
            ctx.n_saved = len(saved)
        ctx.geoms = geoms
        ctx.n_main = blk.n_main
        ctx.has_down = len(blk.unit_specs) > blk.n_main
        ctx.bn_training = training
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.bn_training:
            raise NotImplementedError(_EVAL_BN_BACKWARD)
        t = ctx.saved_tensors
        saved, params = t[:ctx.n_saved], t[ctx.n_saved:]
        need = ctx.needs_input_grad      # (x, blk, training, *params)
        with wgrad_batch():
            dx, grads, _ = _block_backward(saved, params, ctx.geoms, ctx.n_main, ctx.has_down, dout, need[3:], need[0])
        return (dx, None, None, *grads)


class StageLink:
    """Side channel between two consecutive stages run as ResStageFn nodes, valid only while the first stage's output has the
    second stage as its ONLY consumer (no hook on the stage module): the producer leaves the BatchNorm-backward operands of its
    last unit (``stats``), the consumer's backward -- whose first conv1 dgrad writes the gradient w.r.t. that output -- takes their
    statistics in its epilogue and leaves the tile sums (``partial``) for the producer's backward, which runs after it."""
    __slots__ = ('stats', 'partial')

    def __init__(self):
        self.stats = None
        self.partial = None


CROSS_STAGE_STATS = _os.environ.get('BDVCIL_CROSS_STAGE_STATS', '1') != '0'


class ResStageFn(torch.autograd.Function):
    """A whole stage (UPSTREAM ResNet.layerN = a sequence of residual blocks) as ONE autograd node.  The tensors
    between its blocks are then private to this node (exactly one producer and one consumer), which is what allows
    block k+1's conv1 dgrad -- the kernel that writes the gradient w.r.t. block k's output -- to take the
    BatchNorm-backward statistics of block k's last unit in its epilogue instead of a separate pass over that
    (widest) tensor.  Arithmetic per block is that of ResBlockFn."""

    @staticmethod
    def forward(ctx, x, blocks, training, in_link, out_link, *params):
        save = training and any(ctx.needs_input_grad)
        cur, off = x, 0
        all_saved, meta = [], []
        for blk in blocks:
            npar = 3 * len(blk.unit_specs)
            out, saved, geoms = _block_forward(cur, blk, training, params[off:off + npar], save)
            meta.append((len(saved), npar, geoms, blk.n_main, len(blk.unit_specs) > blk.n_main))
            all_saved += saved
            off += npar
            cur = out
        # the previous stage's last unit: operands for the statistics this stage's first conv1 dgrad can take for it
        in_stats = in_link.stats if (save and in_link is not None and CROSS_STAGE_STATS and FUSE_BN_STATS) else None
        if in_stats is not None and not (meta[0][2][0].stride == 1 and ctx.needs_input_grad[0]):
            in_stats = None
        if save:
            ctx.save_for_backward(*all_saved, *params, *(in_stats or ()))
            if out_link is not None:
                out_link.stats = _block_out_stats(all_saved[len(all_saved) - meta[-1][0]:], meta[-1][3])
        ctx.in_link = in_link if in_stats is not None else None
        ctx.out_link = out_link if save else None
        ctx.meta = meta
        ctx.bn_training = training
        return cur

    @staticmethod
    def backward(ctx, dout):
        if not ctx.bn_training:
            raise NotImplementedError(_EVAL_BN_BACKWARD)
        t = ctx.saved_tensors
        meta = ctx.meta
        n_saved_total = sum(m[0] for m in meta)
        in_stats = None
        if ctx.in_link is not None:
            t, in_stats = t[:-4], tuple(t[-4:])
        saved_all, params_all = t[:n_saved_total], t[n_saved_total:]
        need = ctx.needs_input_grad      # (x, blocks, training, in_link, out_link, *params)
        s_off = [0]
        p_off = [0]
        for m in meta:
            s_off.append(s_off[-1] + m[0])
            p_off.append(p_off[-1] + m[1])
        grads_all: List[Optional[torch.Tensor]] = [None] * len(params_all)
        d, part = dout, None
        if ctx.out_link is not None:     # the next stage's backward ran before this one and took the statistics of `dout`
            part, ctx.out_link.partial = ctx.out_link.partial, None
        with wgrad_batch():              # one split-K reduction launch for the weight gradients of the whole stage
            for k in range(len(meta) - 1, -1, -1):
                n_saved, npar, geoms, n_main, has_down = meta[k]
                saved = saved_all[s_off[k]:s_off[k + 1]]
                params = params_all[p_off[k]:p_off[k + 1]]
                prev_stats = in_stats if k == 0 else None
                if k > 0:
                    pm = meta[k - 1]
                    prev_stats = _block_out_stats(saved_all[s_off[k - 1]:s_off[k]], pm[3])
                need_dx = need[0] or k > 0
                d, grads, part = _block_backward(saved, params, geoms, n_main, has_down, d, need[5 + p_off[k]:5 + p_off[k + 1]],
                                                 need_dx, out_stat_partial=part, prev_stats=prev_stats)
                grads_all[p_off[k]:p_off[k + 1]] = grads
        if ctx.in_link is not None:
            ctx.in_link.partial = part   # tile sums of the statistics of `d` against the previous stage's last unit
        return (d, None, None, None, None, *grads_all)


class AvgPoolFn(torch.autograd.Function):
    """UPSTREAM TSMHead.avg_pool = AdaptiveAvgPool2d(1) on an NCHW view of NHWC storage."""

    @staticmethod
    def forward(ctx, x_nchw):
        x = nchw_view_to_nhwc(x_nchw)
        ctx.in_shape = tuple(x.shape)
        ctx.in_dtype = x.dtype          # bf16 storage ends here: the pooled features and everything after them are fp32
        return K.avgpool_fwd(x).view(x.shape[0], x.shape[3], 1, 1)

    @staticmethod
    def backward(ctx, dout):
        N, H, W, C = ctx.in_shape
        d = dout.reshape(N, C)
        d = d if d.is_contiguous() else d.contiguous()
        return nhwc_to_nchw_view(K.avgpool_bwd(d, ctx.in_shape, ctx.in_dtype))


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return K.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dout):
        return K.dropout(dout.contiguous(), ctx.p, ctx.seed), None, None


class LSCFn(torch.autograd.Function):
    """libs/models/cil_heads/cosine_linear.py:27-43."""

    @staticmethod
    def forward(ctx, x, weights, out_features, nb_proxies):
        x = x.contiguous()
        sim, xn, wn, cb = K.lsc_fwd(x, weights, out_features, nb_proxies)
        ctx.save_for_backward(x, weights, xn, wn, cb)
        ctx.kp = (out_features, nb_proxies)
        return sim

    @staticmethod
    def backward(ctx, dsim):
        x, w, xn, wn, cb = ctx.saved_tensors
        Kc, P = ctx.kp
        dx, dw = K.lsc_bwd(dsim.contiguous(), x, w, xn, wn, cb, Kc, P, need_dw=ctx.needs_input_grad[1])
        return dx, dw, None, None


class LinearFn(torch.autograd.Function):
    """libs/models/cil_heads/inc_net.py:36-37."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return K.linear_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx, dw, db = K.linear_bwd(dout.contiguous(), x, w, need_dx=ctx.needs_input_grad[0], need_dw=need_w,
                                  need_db=ctx.has_bias)
        return dx, dw, (db if ctx.has_bias else None)


class ConsensusFn(torch.autograd.Function):
    """UPSTREAM AvgConsensus(dim=1): (B,T,K) -> (B,1,K)."""

    @staticmethod
    def forward(ctx, x):
        B, T, Kc = x.shape
        ctx.T = T
        return K.consensus_fwd(x.reshape(B * T, Kc).contiguous(), B, T).view(B, 1, Kc)

    @staticmethod
    def backward(ctx, dout):
        B, _, Kc = dout.shape
        return K.consensus_bwd(dout.reshape(B, Kc).contiguous(), ctx.T).view(B, ctx.T, Kc)


def _scale_by(grad_out: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Upstream scalar gradient times a small saved tensor ((B,K) or (1,)); host-free."""
    return t * grad_out


class LSCLossFn(torch.autograd.Function):
    """libs/losses/lsc_loss.py:36-56; forward kernel also produces dsim and deta."""

    @staticmethod
    def forward(ctx, sim, targets, eta, margin, hinge, class_weights=None):
        loss, dsim, deta = K.lsc_loss(sim.contiguous(), targets.contiguous(), eta, margin, hinge, class_weights)
        ctx.save_for_backward(dsim, deta)
        return loss

    @staticmethod
    def backward(ctx, g):
        dsim, deta = ctx.saved_tensors
        return _scale_by(g, dsim), None, (_scale_by(g, deta) if ctx.needs_input_grad[2] else None), None, None, None


class SoftCEFn(torch.autograd.Function):
    """libs/cil/icarl.py:123-125 (soft targets) or plain mean cross-entropy (integer labels)."""

    @staticmethod
    def forward(ctx, score, soft_targets, labels):
        loss, dscore = K.softce_loss(score.contiguous(), soft_targets, labels)
        ctx.save_for_backward(dscore)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dscore,) = ctx.saved_tensors
        return _scale_by(g, dscore), None, None


class KDMSEFn(torch.autograd.Function):
    """nn.MSELoss() between hooked feature maps (libs/cil/cil.py:519-541); ``prev`` gets no gradient."""

    @staticmethod
    def forward(ctx, cur, prev):
        ctx.save_for_backward(cur, prev)
        return K.kd_mse_fwd(cur, prev)

    @staticmethod
    def backward(ctx, g):
        cur, prev = ctx.saved_tensors
        return K.kd_mse_bwd(cur, prev, g.reshape(1).contiguous(), 1.0), None


def kd_mse(cur: torch.Tensor, prev: torch.Tensor) -> torch.Tensor:
    return KDMSEFn.apply(cur, prev.detach())
