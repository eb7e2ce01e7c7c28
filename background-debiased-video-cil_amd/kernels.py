"""Tensor-level wrappers over the C ABI: shape/dtype/device checks on the host, then a launch on
torch's current HIP stream.  PyTorch is only the owner of device memory and streams here.

Every wrapper validates that operand shapes match what the kernel grid assumes BEFORE launching
(an out-of-bounds access on the GPU can reset the whole node).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from ._lib import BnStatFuse, ConvAffine, ConvGeom, check, lib

_WS = {}  # (device index, tag) -> workspace tensor (grown on demand, never shrunk)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk(t: torch.Tensor, shape: Optional[Sequence[int]] = None, dtype=torch.float32, name='tensor'):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a tensor, got {type(t)}')
    if not t.is_cuda:
        raise RuntimeError(f'{name}: the HIP path needs a GPU tensor (got device {t.device}); there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name}: dtype {t.dtype}, expected {dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous, got strides {t.stride()} for shape {tuple(t.shape)}')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name}: shape {tuple(t.shape)}, expected {tuple(shape)}')
    return t


ACT_DTYPE = torch.float32       # storage type of activations / their gradients: torch.float32, or torch.bfloat16 ('bf16' arithmetic)


def _act_code(t: torch.Tensor) -> int:
    """BDV_ACT_F32 (0) | BDV_ACT_BF16 (1) of an activation tensor."""
    return 1 if t.dtype == torch.bfloat16 else 0


def _chk_act(t: torch.Tensor, shape=None, name='tensor', like: Optional[torch.Tensor] = None):
    """An activation operand: fp32, or bf16 in the bf16-storage mode; ``like``: must have that tensor's dtype."""
    if isinstance(t, torch.Tensor) and t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f'{name}: dtype {t.dtype}, expected torch.float32 or torch.bfloat16')
    if like is not None and isinstance(t, torch.Tensor) and t.dtype != like.dtype:
        raise TypeError(f'{name}: dtype {t.dtype} differs from the other activation operand ({like.dtype})')
    return _chk(t, shape, dtype=t.dtype if isinstance(t, torch.Tensor) else torch.float32, name=name)


def _chk_conv(t: torch.Tensor, shape, g, name):
    """Operand check of a convolution.  A geometry marked ``frames_view`` (a k x 1 temporal convolution described on the
    [B][T][H*W][C] view) takes the usual [B*T][H][W][C] tensors of the same storage: element count and channel count must
    agree, the kernel only sees the pointer."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16:
        if g.act_dtype != 1:
            raise TypeError(f'{name}: bf16 tensor in an fp32 convolution call (operands of one call share the storage type)')
    elif g.act_dtype != 0:
        raise TypeError(f'{name}: dtype {getattr(t, "dtype", None)} in a bf16-storage convolution call')
    if getattr(g, 'frames_view', False):
        _chk(t, None, dtype=t.dtype, name=name)
        n = 1
        for d in shape:
            n *= d
        if t.numel() != n or t.shape[-1] != shape[-1]:
            raise ValueError(f'{name}: shape {tuple(t.shape)} is not a view of {tuple(shape)}')
        return t
    return _chk(t, shape, dtype=torch.bfloat16 if g.act_dtype == 1 else torch.float32, name=name)


def workspace(nbytes: int, device, tag='ws') -> torch.Tensor:
    key = (torch.device(device).index or 0, tag)
    cur = _WS.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = cur
    return cur


def make_geom(N, H, W, Cin, Cout, R, S, stride, pad, T=1, fold=0, pad_w=None, rt=0, st_t=0) -> ConvGeom:
    """``pad`` pads rows and columns unless ``pad_w`` gives the column padding separately (k x 1 temporal convolutions).
    ``rt`` > 1: the I3D stem's temporal taps (Cin = 4): N = output frames, T = output frames per clip, temporal stride ``st_t``;
    the input then has N * st_t frames and the weight is (Cout, rt * R, S, 4)."""
    pw = pad if pad_w is None else pad_w
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pw - S) // stride + 1
    return ConvGeom(N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, T, fold, -1 if pad_w is None else pad_w, rt, st_t)


def _in_frames(g: ConvGeom) -> int:
    return g.N * g.st_t if g.Rt > 1 else g.N


def _taps_r(g: ConvGeom) -> int:
    return g.R * g.Rt if g.Rt > 1 else g.R


def make_temporal_geom(B, T, H, W, Cin, Cout, kt) -> ConvGeom:
    """k x 1 x 1 temporal convolution (stride 1, padding k // 2) of [B*T][H][W][C] frames, as a k x 1 convolution on the
    [B][T][H*W][C] view of the same storage (I3D Bottleneck3d.conv1 with inflate, configs/_base_/models/i3d_r50.py:13)."""
    g = make_geom(B, T, H * W, Cin, Cout, kt, 1, 1, kt // 2, pad_w=0)
    g.frames_view = True
    return g


# ---------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------

WS_TAG_SUFFIX = ''     # set while launching on another stream: two streams must not share a K-split workspace


def _conv_ws(g: ConvGeom, kind: int, device, tag: str) -> torch.Tensor:
    need = lib().bdv_conv_workspace_bytes(ctypes.byref(g), kind)
    if need == 0:
        check(-1, 'bdv_conv_workspace_bytes')
    return workspace(need, device, tag + WS_TAG_SUFFIX)


import os as _os

# Arithmetic of the 128-wide convolution tiles.  Default: every fp32 product is formed from three
# bf16 pieces per operand (six v_mfma_f32_32x32x16_bf16 products of relative weight >= 2^-16, fp32 accumulate; the dropped
# terms are below 2^-24 relative, the level of fp32 rounding).  BDVCIL_CONV_F32MFMA=1 selects the v_mfma_f32_32x32x2_f32
# kernels (an exact fp32 FMA chain) for all three directions; the per-direction switches override it.
_F32MFMA = _os.environ.get('BDVCIL_CONV_F32MFMA', '0') != '0'


def _x3_default(name):
    v = _os.environ.get(name)
    return (not _F32MFMA) if v is None else v != '0'


FPROP_X3 = _x3_default('BDVCIL_FPROP_X3')


PIECES = 3      # 3: fp32-level products from three bf16 pieces per operand; 2: two pieces, three products ('bf16x2', 16 significand
                # bits); 1: single bf16 product (reduced precision, 'bf16x1')


def set_conv_arith(mode: str):
    """'bf16x3' (default), 'f32mfma', 'bf16x2' (two bf16 pieces per operand, three MFMA products, 16 significand bits; the plane
    kernels only, the other sites keep three pieces), or the reduced-precision modes of BASELINE config 5: 'bf16x1' (operands rounded to bf16, one
    MFMA product, fp32 accumulate, fp32 tensors) and 'bf16' (the same arithmetic with activations and their gradients STORED as
    bf16 between the stem's max-pool and the average pool; fp32 statistics, weights and weight gradients) for fprop, dgrad and
    wgrad at once; returns the previous (fprop, dgrad, wgrad) flags.
    Leaving 'bf16x1' needs another ``set_conv_arith`` call (the flags alone do not restore ``PIECES``)."""
    global FPROP_X3, DGRAD_X3, WGRAD_X3, PIECES, ACT_DTYPE
    if mode not in ('bf16x3', 'f32mfma', 'bf16x2', 'bf16x1', 'bf16'):
        raise ValueError(f"set_conv_arith: unknown mode {mode!r} ('bf16x3' | 'f32mfma' | 'bf16x2' | 'bf16x1' | 'bf16')")
    prev = (FPROP_X3, DGRAD_X3, WGRAD_X3)
    FPROP_X3 = DGRAD_X3 = WGRAD_X3 = (mode != 'f32mfma')
    PIECES = 1 if mode in ('bf16x1', 'bf16') else 2 if mode == 'bf16x2' else 3
    # 'bf16': the bf16x1 arithmetic on bf16-STORED activations and gradients (BDV_ACT_BF16) from the stem's max-pool to the average
    # pool; the tensors the model creates from here on carry the type, every kernel follows the dtype of what it is given
    ACT_DTYPE = torch.bfloat16 if mode == 'bf16' else torch.float32
    return prev


# ---- weights as bf16 planes (bdv_conv_split_weights), cached per weight tensor -----------------------------------------
# A conv weight is cut into its hi / mid / lo planes once per value: the entry of a weight is refreshed when torch's version
# counter of the tensor moved (any in-place torch op, load_state_dict, torch optimizers), when its storage moved, or when
# WEIGHT_EPOCH moved -- the fused SGD kernel writes through raw pointers, which torch cannot see, so FusedSGD.step() calls
# bump_weight_epoch().  Forward and backward of one step share the planes; a frozen teacher's planes live as long as it does.
import weakref as _weakref

WEIGHT_EPOCH = 0
_PLANES = {}        # id(base tensor) -> [weakref, (data_ptr, version, epoch), uint8 buffer holding both layouts, view, geometry, checksum]
# A write that torch's version counter does not see -- ``p.data.mul_()``, ``p.data.copy_()``, ``dist.broadcast(p.data)``, any kernel
# given ``p.data_ptr()`` -- leaves the stamp unchanged and the cached planes STALE: the convolutions would then run on the old
# weights without any error.  Such writers must call ``bump_weight_epoch()`` ("Weights written behind torch's
# back"); the writers inside this package (FusedSGD.step, broadcast_parameters) do.  BDVCIL_CHECK_PLANES=1 is the debugging aid for
# plugin code: every ``weight_planes`` hit then compares a checksum of the weight's bits with the one taken when the planes were
# cut (one device reduction and one host synchronisation per convolution: slow, never for timing) and raises on a mismatch.
CHECK_PLANES = _os.environ.get('BDVCIL_CHECK_PLANES', '0') != '0'


def _weight_checksum(w: torch.Tensor) -> int:
    return int(w.contiguous().view(torch.int32).to(torch.int64).sum().item())
USE_PL = _os.environ.get('BDVCIL_PL', '1') != '0'       # 0: the round-1 bf16-piece kernels (operands split in the K loop)
USE_PL_WGRAD = _os.environ.get('BDVCIL_PL_WGRAD', '1') != '0'


def bump_weight_epoch(params=None):
    """Tell the plane cache that weights were changed behind torch's back (raw-pointer kernels): the given parameters, or
    every cached weight when ``params`` is None (a frozen teacher's planes survive the student's optimizer steps)."""
    global WEIGHT_EPOCH
    if params is None:
        WEIGHT_EPOCH += 1
        return
    for p in params:
        ent = _PLANES.get(id(p))
        if ent is not None:
            ent[1] = None


_PLANES_PENDING = {}    # device index -> (event of the last refresh_weight_planes() on another stream, streams ordered behind it)


def _plane_stamp(w, base):
    return (w.data_ptr(), base._version, WEIGHT_EPOCH)


def weight_planes(w: torch.Tensor, g: ConvGeom):
    """(planes_fprop, planes_dgrad) of the (Cout, R, S, Cin) weight view ``w``: uint8 views of one cached buffer."""
    if _PLANES_PENDING:
        pend = _PLANES_PENDING.get(w.device.index)
        if pend is not None:                            # (event, stream ids that are ordered behind it already)
            cur = torch.cuda.current_stream(w.device)
            if cur.cuda_stream not in pend[1]:
                cur.wait_event(pend[0])
                pend[1].add(cur.cuda_stream)
    base = w._base if w._base is not None else w
    key = id(base)
    stamp = _plane_stamp(w, base)
    ent = _PLANES.get(key)
    if ent is not None and ent[0]() is not base:
        ent = None                                      # the id was recycled by another tensor
    nbytes = lib().bdv_conv_weight_planes_bytes(ctypes.byref(g))
    if nbytes == 0:
        check(-1, 'bdv_conv_weight_planes_bytes')
    if ent is None or ent[1] != stamp or ent[2].numel() != 2 * nbytes:
        buf = ent[2] if ent is not None and ent[2].numel() == 2 * nbytes and ent[2].device == w.device else \
            torch.empty(2 * nbytes, dtype=torch.uint8, device=w.device)
        check(lib().bdv_conv_split_weights(_p(w), ctypes.byref(g), _p(buf[:nbytes]), _p(buf[nbytes:]), _stream()),
              'bdv_conv_split_weights')
        ref = _weakref.ref(base, lambda _r, k=key: _PLANES.pop(k, None))
        # how to rebuild the view and the geometry without holding the parameter alive (refresh_weight_planes)
        ent = [ref, stamp, buf, (tuple(w.shape), tuple(w.stride()), w.storage_offset()), (g.R, g.S, g.Cin, g.Cout),
               _weight_checksum(w) if CHECK_PLANES else None]
        _PLANES[key] = ent
    elif CHECK_PLANES and ent[5] is not None and _weight_checksum(w) != ent[5]:
        raise RuntimeError(
            'stale weight planes: a convolution weight of shape %s was changed without torch noticing (a write through `.data`, a '
            'raw-pointer kernel, dist.broadcast(p.data), ...) after its bf16 planes were cut, so fprop / dgrad would run on the OLD '
            'weights.  Call bdvcil_amd.bump_weight_epoch() after such a write (INTEGRATION.md).' % (tuple(w.shape),))
    return ent[2][:nbytes], ent[2][nbytes:]


def refresh_weight_planes(stream: Optional[torch.cuda.Stream] = None) -> int:
    """Re-split every cached weight whose planes are stale (after an optimizer step) in one go, on ``stream`` (default: the
    current one).  On another stream the work overlaps what the current stream does next (front-end, stem); the first
    ``weight_planes`` call afterwards makes its stream wait for it.  Returns the number of weights re-split."""
    n = 0
    by_dev = {}
    for key, ent in list(_PLANES.items()):
        base = ent[0]()
        if base is None or not base.is_cuda:
            continue
        shape, stride, off = ent[3]
        w = base.as_strided(shape, stride, off) if (tuple(base.shape), tuple(base.stride())) != (shape, stride) else base
        if ent[1] == _plane_stamp(w, base):
            continue
        by_dev.setdefault(base.device, []).append((ent, w, base))
    for device, todo in by_dev.items():
        cur = torch.cuda.current_stream(device)
        st = stream if stream is not None and stream.device == device else cur
        if st != cur:
            st.wait_stream(cur)
        with torch.cuda.stream(st):
            for ent, w, base in todo:
                R, S, Cin, Cout = ent[4]
                g = make_geom(1, R, S, Cin, Cout, R, S, 1, 0, 1, 0)
                buf = ent[2]
                nbytes = buf.numel() // 2
                check(lib().bdv_conv_split_weights(_p(w), ctypes.byref(g), _p(buf[:nbytes]), _p(buf[nbytes:]), _stream()),
                      'bdv_conv_split_weights')
                ent[1] = _plane_stamp(w, base)
                if CHECK_PLANES:
                    ent[5] = _weight_checksum(w)
                n += 1
            if st != cur:
                ev = torch.cuda.Event()
                ev.record(st)
                _PLANES_PENDING[device.index] = (ev, {st.cuda_stream})
    return n


def conv_kernel_name(g: ConvGeom, kind: str, x3: Optional[bool] = None) -> str:
    """Name of the main kernel ``conv_fprop`` / ``conv_dgrad`` / ``conv_wgrad(_partial)`` launches for this geometry."""
    k = {'fprop': 0, 'dgrad': 1, 'wgrad': 2}[kind]
    flag = (FPROP_X3, DGRAD_X3, WGRAD_X3)[k] if x3 is None else x3
    buf = ctypes.create_string_buffer(128)
    check(lib().bdv_conv_kernel_name(ctypes.byref(g), k, {3: 1, 1: 2, 2: 3}[PIECES] if flag else 0, buf, 128), 'bdv_conv_kernel_name')
    return buf.value.decode()


def conv_uses_planes(g: ConvGeom, kind: str) -> bool:
    """True when ``conv_fprop`` / ``conv_dgrad`` of this geometry read the cached bf16 weight planes (the cache that
    ``bump_weight_epoch`` invalidates) in the current arithmetic."""
    k = {'fprop': 0, 'dgrad': 1}[kind]
    flag = (FPROP_X3, DGRAD_X3)[k]
    return bool(flag and USE_PL and lib().bdv_conv_uses_planes(ctypes.byref(g), k, PIECES))


PRE_BN_1X1_ONLY = _os.environ.get('BDVCIL_PRE_BN_3X3', '0') == '0'


def fprop_pre_ok(g: ConvGeom) -> bool:
    """True when ``conv_fprop(pre_bn=...)`` and ``conv_wgrad*(pre_bn=...)`` can apply the producer's BatchNorm + ReLU in their
    loaders for this geometry (bf16-piece arithmetic with the plane kernels; no temporal shift)."""
    if PRE_BN_1X1_ONLY and (g.R != 1 or g.S != 1):   # a 3x3 consumer re-applies the BatchNorm once per filter tap: measured slower
        return False
    return bool(PIECES == 3 and FPROP_X3 and WGRAD_X3 and DGRAD_X3 and USE_PL and USE_PL_WGRAD and g.fold == 0 and g.Rt <= 1
                and not getattr(g, 'frames_view', False)
                and lib().bdv_conv_fprop_pre_ok(ctypes.byref(g)) and lib().bdv_conv_wgrad_pre_ok(ctypes.byref(g)))


def conv_fprop(x: torch.Tensor, w: torch.Tensor, g: ConvGeom, out: Optional[torch.Tensor] = None,
               ws_tag: str = 'conv', bn_stats: bool = False, affine=None, x3: Optional[bool] = None, pre_bn=None):
    """x (N,H,W,Cin) NHWC, w (Cout,R,S,Cin) -> y (N,Ho,Wo,Cout); with bn_stats also the fused BatchNorm partial
    sums (float[2][rows][Cout]) for ``bn_train_finalize``.
    ``affine = (scale, shift, residual | None, relu)``: eval-mode BatchNorm folded into the epilogue,
    y = relu?(conv * scale + shift (+ residual)).
    ``pre_bn = (scale, shift)``: x is the RAW output of the producing conv; its train-mode BatchNorm + ReLU is applied in the
    loader (``fprop_pre_ok(g)``), so no apply pass / activation / mask is needed for this consumer."""
    g.act_dtype = _act_code(x)
    _chk_conv(x, (_in_frames(g), g.H, g.W, g.Cin), g, 'x')
    _chk(w, (g.Cout, _taps_r(g), g.S, g.Cin), name='w')
    y = out if out is not None else torch.empty((g.N, g.Ho, g.Wo, g.Cout), dtype=x.dtype, device=x.device)
    _chk_conv(y, (g.N, g.Ho, g.Wo, g.Cout), g, 'y')
    ws = _conv_ws(g, 0, x.device, ws_tag)
    part = aff = None
    use_x3 = FPROP_X3 if x3 is None else x3
    use_pl = bool(use_x3 and USE_PL and lib().bdv_conv_uses_planes(ctypes.byref(g), 0, PIECES))
    if g.act_dtype == 1 and not (use_pl and PIECES == 1):
        raise ValueError("conv_fprop: bf16 tensors need set_conv_arith('bf16') and a geometry of the plane kernels (Cin % 32 == 0, Cout % 64 == 0)")
    if pre_bn is not None:
        if not (use_x3 and USE_PL and PIECES == 3 and lib().bdv_conv_fprop_pre_ok(ctypes.byref(g))):
            raise ValueError('conv_fprop: pre_bn needs the bf16-piece plane kernels for this geometry (fprop_pre_ok)')
        _chk(pre_bn[0], (g.Cin,), name='pre_scale')
        _chk(pre_bn[1], (g.Cin,), name='pre_shift')
        use_pl = True
    if bn_stats:
        if affine is not None:
            raise ValueError('conv_fprop: bn_stats and affine exclude each other')
        rows = lib().bdv_conv_fprop_pre_stat_rows(ctypes.byref(g)) if pre_bn is not None else \
            lib().bdv_conv_fprop_pl_stat_rows(ctypes.byref(g), PIECES) if use_pl else lib().bdv_conv_fprop_stat_rows(ctypes.byref(g))
        if rows <= 0:
            check(-1, 'bdv_conv_fprop_stat_rows')
        part = torch.empty((2, rows, g.Cout), dtype=torch.float32, device=x.device)
    if affine is not None:
        scale, shift, res, relu = affine
        _chk(scale, (g.Cout,), name='scale')
        _chk(shift, (g.Cout,), name='shift')
        if res is not None:
            _chk_conv(res, (g.N, g.Ho, g.Wo, g.Cout), g, 'residual')
        aff = ConvAffine(scale.data_ptr(), shift.data_ptr(), res.data_ptr() if res is not None else None, int(bool(relu)))
    if use_pl:
        planes_f, _ = weight_planes(w, g)
        check(lib().bdv_conv_fprop_pl(_p(x), _p(w), _p(planes_f), _p(y), ctypes.byref(g), _p(part),
                                      ctypes.byref(aff) if aff is not None else None, _p(ws), ws.numel(), PIECES,
                                      _p(pre_bn[0] if pre_bn is not None else None), _p(pre_bn[1] if pre_bn is not None else None),
                                      _stream()), 'bdv_conv_fprop_pl')
        return (y, part) if bn_stats else y
    fn = lib().bdv_conv_fprop_x3 if use_x3 else lib().bdv_conv_fprop
    check(fn(_p(x), _p(w), _p(y), ctypes.byref(g), _p(part), ctypes.byref(aff) if aff is not None else None,
             _p(ws), ws.numel(), _stream()), 'bdv_conv_fprop')
    return (y, part) if bn_stats else y


DGRAD_X3 = _x3_default('BDVCIL_DGRAD_X3')


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, g: ConvGeom, add_src: Optional[torch.Tensor] = None,
               add_mask_src: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               ws_tag: str = 'conv', bn_stats=None, x3: Optional[bool] = None):
    """``bn_stats = (y, relu_mask | None, mean, invstd)`` of the conv unit whose output gradient this dgrad produces:
    the BatchNorm-backward statistics are then taken in the epilogue and ``(dx, partial)`` is returned; pass ``partial``
    to ``bn_backward(stat_partial=...)``.  Stride 2 needs a filter that reaches every input pixel (R, S >= 2)."""
    g.act_dtype = _act_code(dy)
    _chk_conv(dy, (g.N, g.Ho, g.Wo, g.Cout), g, 'dy')
    _chk(w, (g.Cout, g.R, g.S, g.Cin), name='w')
    dx = out if out is not None else torch.empty((g.N, g.H, g.W, g.Cin), dtype=dy.dtype, device=dy.device)
    _chk_conv(dx, (g.N, g.H, g.W, g.Cin), g, 'dx')
    if add_src is not None:
        _chk_conv(add_src, (g.N, g.H, g.W, g.Cin), g, 'add_src')
        if add_src.data_ptr() == dx.data_ptr() and g.fold > 0:
            raise ValueError('conv_dgrad: in-place add_src is not allowed with a temporal shift (scatter epilogue)')
    if add_mask_src is not None:
        _chk(add_mask_src, (g.N * g.H * g.W * g.Cin // 32,), dtype=torch.int32, name='add_mask_src')
    fuse = partial = None
    use_x3 = (DGRAD_X3 if x3 is None else x3) and g.Cin % 64 == 0
    use_pl = bool(use_x3 and USE_PL and lib().bdv_conv_uses_planes(ctypes.byref(g), 1, PIECES))
    if g.act_dtype == 1 and not (use_pl and PIECES == 1):
        raise ValueError("conv_dgrad: bf16 tensors need set_conv_arith('bf16') and a geometry of the plane kernels (Cout % 32 == 0, Cin % 64 == 0)")
    relu_affine = None
    if bn_stats is not None:
        if len(bn_stats) == 5:      # (y, None, mean, invstd, (scale, shift)): the unit's ReLU sign is derived from y
            y, mask, mean, invstd, relu_affine = bn_stats
        else:
            y, mask, mean, invstd = bn_stats
        _chk_conv(y, (g.N, g.H, g.W, g.Cin), g, 'y')
        _chk(mean, (g.Cin,), name='mean')
        _chk(invstd, (g.Cin,), name='invstd')
        if mask is not None:
            _chk(mask, (y.numel() // 32,), dtype=torch.int32, name='relu_mask')
        rows = lib().bdv_conv_dgrad_pl_stat_rows(ctypes.byref(g), PIECES) if use_pl else lib().bdv_conv_dgrad_stat_rows(ctypes.byref(g))
        if rows <= 0:
            check(-1, 'bdv_conv_dgrad_stat_rows')
        partial = torch.empty((2, rows, g.Cin), dtype=torch.float32, device=dy.device)
        if relu_affine is not None:
            if mask is not None:
                raise ValueError('conv_dgrad: a ReLU mask and a derived ReLU sign exclude each other')
            _chk(relu_affine[0], (g.Cin,), name='relu_scale')
            _chk(relu_affine[1], (g.Cin,), name='relu_shift')
        fuse = BnStatFuse(y.data_ptr(), mask.data_ptr() if mask is not None else None, mean.data_ptr(), invstd.data_ptr(),
                          partial.data_ptr(), relu_affine[0].data_ptr() if relu_affine is not None else None,
                          relu_affine[1].data_ptr() if relu_affine is not None else None)
    ws = _conv_ws(g, 1, dy.device, ws_tag)
    if use_pl:
        _, planes_d = weight_planes(w, g)
        check(lib().bdv_conv_dgrad_pl(_p(dy), _p(w), _p(planes_d), _p(dx), _p(add_src), _p(add_mask_src), ctypes.byref(g),
                                      ctypes.byref(fuse) if fuse is not None else None, _p(ws), ws.numel(), PIECES, _stream()),
              'bdv_conv_dgrad_pl')
        return dx if bn_stats is None else (dx, partial)
    if use_x3 and g.Cin % 128 == 0:
        w_t = w.permute(1, 2, 3, 0).contiguous()            # (R, S, Cin, Cout): contraction index contiguous
        check(lib().bdv_conv_dgrad_x3(_p(dy), _p(w), _p(w_t), _p(dx), _p(add_src), _p(add_mask_src), ctypes.byref(g),
                                      ctypes.byref(fuse) if fuse is not None else None, _p(ws), ws.numel(), _stream()),
              'bdv_conv_dgrad_x3')
        return dx if bn_stats is None else (dx, partial)
    check(lib().bdv_conv_dgrad(_p(dy), _p(w), _p(dx), _p(add_src), _p(add_mask_src), ctypes.byref(g),
                               ctypes.byref(fuse) if fuse is not None else None, _p(ws), ws.numel(), _stream()), 'bdv_conv_dgrad')
    return dx if bn_stats is None else (dx, partial)


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, g: ConvGeom, dw: Optional[torch.Tensor] = None,
               beta: float = 0.0, ws_tag: str = 'wgrad', x3: Optional[bool] = None, pre_bn=None) -> torch.Tensor:
    """dw = beta * dw + dy^T (*) x in one call (split-K main kernel + fixed-order reduction)."""
    g.act_dtype = _act_code(dy)
    _chk_conv(dy, (g.N, g.Ho, g.Wo, g.Cout), g, 'dy')
    _chk_conv(x, (_in_frames(g), g.H, g.W, g.Cin), g, 'x')
    if dw is None:
        dw = torch.empty((g.Cout, _taps_r(g), g.S, g.Cin), dtype=torch.float32, device=dy.device)
        beta = 0.0
    _chk(dw, (g.Cout, _taps_r(g), g.S, g.Cin), name='dw')
    if WGRAD_X3 if x3 is None else x3:      # the bf16-piece main kernel only exists in the partial + reduce form
        slab, _ = conv_wgrad_partial(dy, x, g, x3=True, dw=dw, pre_bn=pre_bn)
        wgrad_reduce_batched([(slab, dw)], beta=beta)
        return dw
    if pre_bn is not None:
        raise ValueError('conv_wgrad: pre_bn needs the bf16-piece plane kernel')
    ws = _conv_ws(g, 2, dy.device, ws_tag)
    check(lib().bdv_conv_wgrad(_p(dy), _p(x), _p(dw), float(beta), ctypes.byref(g), _p(ws), ws.numel(), _stream()),
          'bdv_conv_wgrad')
    return dw


# ---------------------------------------------------------------------------------------------
# batch norm (+ReLU, +residual)
# ---------------------------------------------------------------------------------------------

def _bn_ws(M, C, device):
    return workspace(lib().bdv_bn_workspace_bytes(int(M), int(C)), device, 'bn')


def bn_train_stats(y, gamma, beta, eps, momentum, running_mean, running_var):
    """y (..., C) -> (save_mean, save_invstd, scale, shift); running stats updated in place."""
    C = y.shape[-1]
    M = y.numel() // C
    _chk(y, name='y')
    for t, n in ((gamma, 'gamma'), (beta, 'beta')):
        _chk(t, (C,), name=n)
    if running_mean is not None:
        _chk(running_mean, (C,), name='running_mean')
        _chk(running_var, (C,), name='running_var')
    stats = torch.empty((4, C), dtype=torch.float32, device=y.device)
    ws = _bn_ws(M, C, y.device)
    check(lib().bdv_bn_train_stats(_p(y), M, C, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                                   _p(running_var), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), _p(ws),
                                   ws.numel(), _stream()), 'bdv_bn_train_stats')
    return stats[0], stats[1], stats[2], stats[3]


def bn_train_finalize(partial, M, gamma, beta, eps, momentum, running_mean, running_var):
    """partial float[2][rows][C] from conv_fprop(bn_stats=True) -> (save_mean, save_invstd, scale, shift)."""
    _chk(partial, name='partial')
    _, rows, C = partial.shape
    for t, n in ((gamma, 'gamma'), (beta, 'beta')):
        _chk(t, (C,), name=n)
    if running_mean is not None:
        _chk(running_mean, (C,), name='running_mean')
        _chk(running_var, (C,), name='running_var')
    stats = torch.empty((4, C), dtype=torch.float32, device=partial.device)
    check(lib().bdv_bn_train_finalize(_p(partial), rows, int(M), C, _p(gamma), _p(beta), float(eps), float(momentum),
                                      _p(running_mean), _p(running_var), _p(stats[0]), _p(stats[1]), _p(stats[2]),
                                      _p(stats[3]), _stream()), 'bdv_bn_train_finalize')
    return stats[0], stats[1], stats[2], stats[3]


def bn_eval_params(gamma, beta, running_mean, running_var, eps):
    C = gamma.numel()
    for t, n in ((gamma, 'gamma'), (beta, 'beta'), (running_mean, 'running_mean'), (running_var, 'running_var')):
        _chk(t, (C,), name=n)
    ss = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    check(lib().bdv_bn_eval_params(C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(ss[0]),
                                   _p(ss[1]), _stream()), 'bdv_bn_eval_params')
    return ss[0], ss[1]


def bn_apply(y, scale, shift, res=None, relu=True, out=None, want_mask=False, res_affine=None):
    """-> out, or (out, relu_mask) when want_mask (int32 tensor, 1 bit per element: out > 0).
    ``res_affine = (scale, shift)``: the residual is a raw conv output and gets its own BatchNorm here."""
    C = y.shape[-1]
    M = y.numel() // C
    _chk_act(y, name='y')
    _chk(scale, (C,), name='scale')
    _chk(shift, (C,), name='shift')
    if res is not None:
        _chk_act(res, tuple(y.shape), name='res', like=y)
    rs = rb = None
    if res_affine is not None:
        if res is None:
            raise ValueError('bn_apply: res_affine needs res')
        rs, rb = res_affine
        _chk(rs, (C,), name='res_scale')
        _chk(rb, (C,), name='res_shift')
    o = out if out is not None else torch.empty_like(y)
    _chk_act(o, tuple(y.shape), name='out', like=y)
    mask = None
    if want_mask:
        if not relu or C % 32 != 0:
            raise ValueError('bn_apply: a ReLU mask needs relu=True and C % 32 == 0')
        mask = torch.empty(y.numel() // 32, dtype=torch.int32, device=y.device)
    check(lib().bdv_bn_apply(_p(y), _p(scale), _p(shift), _p(res), _p(rs), _p(rb), _p(o), _p(mask), M, C, int(bool(relu)),
                             _act_code(y), _stream()), 'bdv_bn_apply')
    return (o, mask) if want_mask else o


def bn_backward(dout, relu_mask, y, gamma, save_mean, save_invstd, relu, dgamma=None, dbeta=None, beta_acc=0.0, dy=None,
                stat_partial=None, relu_affine=None):
    """Returns (dy, dgamma, dbeta).  ``relu_mask`` is the bit mask from ``bn_apply(want_mask=True)`` (needed when relu), or None
    with ``relu_affine = (scale, shift)``: the sign is then derived from y (a unit whose apply pass never ran).
    ``stat_partial``: the ``(2, rows, C)`` tile sums from ``conv_dgrad(bn_stats=...)``; the statistics pass is skipped."""
    C = y.shape[-1]
    M = y.numel() // C
    _chk_act(y, name='y')
    _chk_act(dout, tuple(y.shape), name='dout', like=y)
    if relu and relu_affine is None:
        _chk(relu_mask, (y.numel() // 32,), dtype=torch.int32, name='relu_mask')
    if relu_affine is not None:
        if not relu or relu_mask is not None:
            raise ValueError('bn_backward: relu_affine goes with relu=True and relu_mask=None')
        _chk(relu_affine[0], (C,), name='relu_scale')
        _chk(relu_affine[1], (C,), name='relu_shift')
    for t, n in ((gamma, 'gamma'), (save_mean, 'save_mean'), (save_invstd, 'save_invstd')):
        _chk(t, (C,), name=n)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
        beta_acc = 0.0
    _chk(dgamma, (C,), name='dgamma')
    _chk(dbeta, (C,), name='dbeta')
    d = dy if dy is not None else torch.empty_like(y)
    _chk_act(d, tuple(y.shape), name='dy', like=y)
    srows = 0
    if stat_partial is not None:
        _chk(stat_partial, name='stat_partial')
        if stat_partial.dim() != 3 or stat_partial.shape[0] != 2 or stat_partial.shape[2] != C:
            raise ValueError(f'bn_backward: stat_partial {tuple(stat_partial.shape)} is not (2, rows, {C})')
        srows = stat_partial.shape[1]
    ws = _bn_ws(M, C, y.device)
    check(lib().bdv_bn_backward(_p(dout), _p(relu_mask if relu else None), _p(y), _p(gamma), _p(save_mean), _p(save_invstd),
                                _p(d), _p(dgamma), _p(dbeta), float(beta_acc), M, C, int(bool(relu)), _p(stat_partial), srows,
                                _p(relu_affine[0] if relu_affine is not None else None),
                                _p(relu_affine[1] if relu_affine is not None else None), _p(ws), ws.numel(), _act_code(y), _stream()), 'bdv_bn_backward')
    return d, dgamma, dbeta


def bn_backward_maxpool(dpool, pool_idx, relu_mask, y, gamma, save_mean, save_invstd):
    """BN(+ReLU) backward behind MaxPool2d(3,2,1) (the stem): the pooled gradient is expanded on the fly.
    -> (dy, dgamma, dbeta)."""
    _chk(y, name='y')
    N, H, W, C = y.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    _chk_act(dpool, (N, Ho, Wo, C), name='dpool')      # the pooled tensor's gradient carries the activation storage type; y / dy are fp32
    _chk(pool_idx, (N, Ho, Wo, C), dtype=torch.uint8, name='pool_idx')
    _chk(relu_mask, (y.numel() // 32,), dtype=torch.int32, name='relu_mask')
    for t, n in ((gamma, 'gamma'), (save_mean, 'save_mean'), (save_invstd, 'save_invstd')):
        _chk(t, (C,), name=n)
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = _bn_ws(N * H * W, C, y.device)
    check(lib().bdv_bn_backward_maxpool(_p(dpool), _p(pool_idx), _p(relu_mask), _p(y), _p(gamma), _p(save_mean), _p(save_invstd),
                                        _p(dy), _p(dgamma), _p(dbeta), 0.0, N, H, W, C, _p(ws), ws.numel(), _act_code(dpool), _stream()),
          'bdv_bn_backward_maxpool')
    return dy, dgamma, dbeta


def relu_bwd(dout, relu_mask, add=None, g=None):
    _chk_act(dout, name='dout')
    _chk(relu_mask, (dout.numel() // 32,), dtype=torch.int32, name='relu_mask')
    if add is not None:
        _chk_act(add, tuple(dout.shape), name='add', like=dout)
    r = g if g is not None else torch.empty_like(dout)
    _chk_act(r, tuple(dout.shape), name='g', like=dout)
    check(lib().bdv_relu_bwd(_p(dout), _p(relu_mask), _p(add), _p(r), dout.numel(), _act_code(dout), _stream()), 'bdv_relu_bwd')
    return r


def add(a, b, out=None):
    _chk_act(a, name='a')
    _chk_act(b, tuple(a.shape), name='b', like=a)
    o = out if out is not None else torch.empty_like(a)
    _chk_act(o, tuple(a.shape), name='out', like=a)
    check(lib().bdv_add(_p(a), _p(b), _p(o), a.numel(), _act_code(a), _stream()), 'bdv_add')
    return o


# ---------------------------------------------------------------------------------------------
# stem helpers / pooling / front-end
# ---------------------------------------------------------------------------------------------

def nchw3_to_nhwc4(x):
    """(N,3,H,W) -> (N,H,W,4)."""
    _chk(x, name='x')
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f'nchw3_to_nhwc4: expected (N,3,H,W), got {tuple(x.shape)}')
    N, _, H, W = x.shape
    out = torch.empty((N, H, W, 4), dtype=torch.float32, device=x.device)
    check(lib().bdv_nchw3_to_nhwc4(_p(x), _p(out), N, H, W, _stream()), 'bdv_nchw3_to_nhwc4')
    return out


def maxpool_fwd(x, out_dtype=torch.float32):
    """``out_dtype``: ``ACT_DTYPE`` at the 2-D stem (the pooled tensor is the first one in the activation storage type)."""
    _chk(x, name='x')
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, C), dtype=out_dtype, device=x.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    check(lib().bdv_maxpool_fwd(_p(x), _p(out), _p(idx), N, H, W, C, _act_code(out), _stream()), 'bdv_maxpool_fwd')
    return out, idx


def bn_relu_maxpool_fwd(y, scale, shift, out_dtype=torch.float32):
    """Stem tail: relu(y * scale + shift) -> MaxPool2d(3,2,1) in one pass -> (pooled, idx, relu_mask of the activation)."""
    _chk(y, name='y')
    N, H, W, C = y.shape
    _chk(scale, (C,), name='scale')
    _chk(shift, (C,), name='shift')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, C), dtype=out_dtype, device=y.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=y.device)
    mask = torch.empty(y.numel() // 32, dtype=torch.int32, device=y.device)
    check(lib().bdv_bn_relu_maxpool_fwd(_p(y), _p(scale), _p(shift), _p(out), _p(idx), _p(mask), N, H, W, C, _act_code(out), _stream()),
          'bdv_bn_relu_maxpool_fwd')
    return out, idx, mask


def maxpool_t2_fwd(x):
    """MaxPool3d((2,1,1),(2,1,1)) on frames (2n, H, W, C) -> ((n, H, W, C), sel bit mask)."""
    _chk(x, name='x')
    N2, H, W, C = x.shape
    if N2 % 2 or (H * W * C) % 32:
        raise ValueError(f'maxpool_t2_fwd: {tuple(x.shape)}: needs an even frame count and H*W*C % 32 == 0')
    out = torch.empty((N2 // 2, H, W, C), dtype=torch.float32, device=x.device)
    sel = torch.empty(out.numel() // 32, dtype=torch.int32, device=x.device)
    check(lib().bdv_maxpool_t2_fwd(_p(x), _p(out), _p(sel), N2 // 2, H * W * C, _stream()), 'bdv_maxpool_t2_fwd')
    return out, sel


def maxpool_t2_bwd(dout, sel):
    _chk(dout, name='dout')
    n, H, W, C = dout.shape
    _chk(sel, (dout.numel() // 32,), dtype=torch.int32, name='sel')
    dx = torch.empty((2 * n, H, W, C), dtype=torch.float32, device=dout.device)
    check(lib().bdv_maxpool_t2_bwd(_p(dout), _p(sel), _p(dx), n, H * W * C, _stream()), 'bdv_maxpool_t2_bwd')
    return dx


def maxpool_bwd(dout, idx, in_shape):
    N, H, W, C = in_shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    _chk_act(dout, (N, Ho, Wo, C), name='dout')
    _chk(idx, (N, Ho, Wo, C), dtype=torch.uint8, name='idx')
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dout.device)
    check(lib().bdv_maxpool_bwd(_p(dout), _p(idx), _p(dx), N, H, W, C, _act_code(dout), _stream()), 'bdv_maxpool_bwd')
    return dx


def avgpool_fwd(x):
    _chk_act(x, name='x')
    N, H, W, C = x.shape
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(lib().bdv_avgpool_fwd(_p(x), _p(out), N, H * W, C, _act_code(x), _stream()), 'bdv_avgpool_fwd')
    return out


def avgpool_bwd(dout, in_shape, dtype=torch.float32):
    """``dtype``: storage type of the pooled tensor's input (its gradient is written in it)."""
    N, H, W, C = in_shape
    _chk(dout, (N, C), name='dout')
    dx = torch.empty((N, H, W, C), dtype=dtype, device=dout.device)
    check(lib().bdv_avgpool_bwd(_p(dout), _p(dx), N, H * W, C, _act_code(dx), _stream()), 'bdv_avgpool_bwd')
    return dx


def bgmix_normalize_u8(frames, bg, mix, alpha, mean, std, want_nhwc4=True, want_nchw=False):
    """frames (B,T,H,W,3) u8, bg (B,H,W,3) u8 -- or fp32 pixel values in [0,255], the output of ``bg_resize_crop_u8`` -- | None,
    mix (B,) u8/bool | None."""
    _chk(frames, dtype=torch.uint8, name='frames')
    B, T, H, W, c3 = frames.shape
    if c3 != 3:
        raise ValueError('frames must be (B,T,H,W,3)')
    bg_f32 = bg is not None and bg.dtype == torch.float32
    if bg is not None:
        _chk(bg, (B, H, W, 3), dtype=torch.float32 if bg_f32 else torch.uint8, name='bg')
        mix = mix.to(torch.uint8) if mix.dtype != torch.uint8 else mix
        _chk(mix, (B,), dtype=torch.uint8, name='mix')
    else:
        mix = None
    f3 = ctypes.c_float * 3
    m = torch.tensor(list(mean), dtype=torch.float32)
    s = torch.tensor(list(std), dtype=torch.float32)
    inv = 1.0 / s      # fp32 reciprocal, as the CPU restatement computes it
    o4 = torch.empty((B * T, H, W, 4), dtype=torch.float32, device=frames.device) if want_nhwc4 else None
    oc = torch.empty((B, T, 3, H, W), dtype=torch.float32, device=frames.device) if want_nchw else None
    check(lib().bdv_bgmix_normalize_u8(_p(frames), _p(bg), int(bg_f32), _p(mix), float(alpha), f3(*m.tolist()), f3(*s.tolist()),
                                       f3(*inv.tolist()), _p(o4), _p(oc), B, T, H, W, _stream()), 'bdv_bgmix_normalize_u8')
    return o4, oc


def resized_size(h: int, w: int, size: int):
    """Output size of torchvision's ``Resize(size)`` with an int: the smaller edge becomes ``size``, the other one
    ``int(size * long / short)`` (UPSTREAM torchvision ``_compute_resized_output_size``)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def bg_resize_crop_u8(bg, size, crop_h, crop_w, top, left):
    """bg (B,Hs,Ws,3) u8 -> (B,crop_h,crop_w,3) fp32 in [0,255]: ``Resize(size)`` + crop at (top[b], left[b]) of
    BackgroundMixDataset.bg_pipeline (libs/loader/comix_loader.py:72-73); top / left: (B,) int32 tensors on the device."""
    _chk(bg, dtype=torch.uint8, name='bg')
    if bg.dim() != 4 or bg.shape[-1] != 3:
        raise ValueError(f'bg_resize_crop_u8: expected (B,Hs,Ws,3), got {tuple(bg.shape)}')
    B, Hs, Ws, _ = bg.shape
    Hr, Wr = resized_size(Hs, Ws, int(size))
    if crop_h > Hr or crop_w > Wr:
        raise ValueError(f'Required crop size {(crop_h, crop_w)} is larger than input image size {(Hr, Wr)}')     # torchvision's message
    _chk(top, (B,), dtype=torch.int32, name='top')
    _chk(left, (B,), dtype=torch.int32, name='left')
    out = torch.empty((B, crop_h, crop_w, 3), dtype=torch.float32, device=bg.device)
    check(lib().bdv_bg_resize_crop_u8(_p(bg), B, Hs, Ws, Hr, Wr, _p(top), _p(left), int(crop_h), int(crop_w), _p(out), _stream()),
          'bdv_bg_resize_crop_u8')
    return out


def resize_linear_u8(frames, Hd: int, Wd: int, boxes=None):
    """``cv2.resize(..., INTER_LINEAR)`` of uint8 frames (.., Hs, Ws, 3) -> (.., Hd, Wd, 3).  ``boxes``: None, or for frames of shape
    (B, T, Hs, Ws, 3) a list / (B, 4) array of per-clip crops (x0, y0, w, h) that are cut out and resized in the same pass
    (MultiScaleCrop + Resize)."""
    _chk(frames, dtype=torch.uint8, name='frames')
    if frames.dim() < 3 or frames.shape[-1] != 3:
        raise ValueError(f'resize_linear_u8: expected (..., H, W, 3), got {tuple(frames.shape)}')
    Hs, Ws = int(frames.shape[-3]), int(frames.shape[-2])
    lead = tuple(frames.shape[:-3])
    N = 1
    for d in lead:
        N *= int(d)
    out = torch.empty(lead + (int(Hd), int(Wd), 3), dtype=torch.uint8, device=frames.device)
    dev_boxes, host_boxes, per = None, None, 1
    if boxes is not None:
        if frames.dim() != 5:
            raise ValueError('resize_linear_u8: per-clip boxes need frames of shape (B, T, H, W, 3)')
        hb = torch.as_tensor(boxes, dtype=torch.int32).reshape(-1, 4).contiguous()
        if hb.shape[0] != frames.shape[0]:
            raise ValueError(f'resize_linear_u8: {hb.shape[0]} boxes for {frames.shape[0]} clips')
        host_boxes, dev_boxes, per = hb, hb.to(frames.device), int(frames.shape[1])
    check(lib().bdv_resize_linear_u8(_p(frames), N, Hs, Ws, _p(dev_boxes), per, host_boxes.data_ptr() if host_boxes is not None else None,
                                     _p(out), int(Hd), int(Wd), _stream()), 'bdv_resize_linear_u8')
    return out


def crop_normalize_u8(frames, crops, crop_h, crop_w, mean, std, want_nhwc4=True, want_nchw=False):
    """frames (B,T,H,W,3) u8; crops = [(x_offset, y_offset, flip), ...] -> (o4 (B*n*T, ch, cw, 4) | None,
    oc (B, n*T, 3, ch, cw) | None), crop-major frame order."""
    _chk(frames, dtype=torch.uint8, name='frames')
    if frames.dim() != 5 or frames.shape[-1] != 3:
        raise ValueError('frames must be (B,T,H,W,3)')
    B, T, H, W, _ = frames.shape
    n = len(crops)
    table = (ctypes.c_int32 * (3 * n))(*[int(v) for c in crops for v in c])
    f3 = ctypes.c_float * 3
    m = torch.tensor(list(mean), dtype=torch.float32)
    inv = 1.0 / torch.tensor(list(std), dtype=torch.float32)      # fp32 reciprocal, as in bgmix_normalize_u8
    o4 = torch.empty((B * n * T, crop_h, crop_w, 4), dtype=torch.float32, device=frames.device) if want_nhwc4 else None
    oc = torch.empty((B, n * T, 3, crop_h, crop_w), dtype=torch.float32, device=frames.device) if want_nchw else None
    check(lib().bdv_crop_normalize_u8(_p(frames), ctypes.cast(table, ctypes.c_void_p), n, int(crop_h), int(crop_w),
                                      f3(*m.tolist()), f3(*inv.tolist()), _p(o4), _p(oc), B, T, H, W, _stream()),
          'bdv_crop_normalize_u8')
    return o4, oc


WGRAD_X3 = _x3_default('BDVCIL_WGRAD_X3')


def conv_wgrad_partial(dy: torch.Tensor, x: torch.Tensor, g: ConvGeom, x3: Optional[bool] = None,
                       dw: Optional[torch.Tensor] = None, pre_bn=None):
    """Split-K partial products of a weight gradient -> (slab (splits, Cout, R, S, Cin), empty dw); reduce them later with
    ``wgrad_reduce_batched`` (the slab must stay alive until then)."""
    g.act_dtype = _act_code(dy)
    _chk_conv(dy, (g.N, g.Ho, g.Wo, g.Cout), g, 'dy')
    _chk_conv(x, (_in_frames(g), g.H, g.W, g.Cin), g, 'x')
    use_x3 = WGRAD_X3 if x3 is None else x3
    use_pl = use_x3 and USE_PL_WGRAD
    if g.act_dtype == 1 and not (use_pl and PIECES == 1 and pre_bn is None):
        raise ValueError("conv_wgrad_partial: bf16 tensors need set_conv_arith('bf16') and the plane kernel")
    if pre_bn is not None:      # x = the producer's raw conv output; its BatchNorm + ReLU is applied in the loader
        if not (use_pl and lib().bdv_conv_wgrad_pre_ok(ctypes.byref(g))):
            raise ValueError('conv_wgrad_partial: pre_bn needs the bf16-piece plane kernel for this geometry')
        _chk(pre_bn[0], (g.Cin,), name='pre_scale')
        _chk(pre_bn[1], (g.Cin,), name='pre_shift')
    splits = (lib().bdv_conv_wgrad_pl_splits if use_pl else lib().bdv_conv_wgrad_splits)(ctypes.byref(g))
    if splits <= 0:
        check(-1, 'bdv_conv_wgrad_splits')
    slab = torch.empty((splits, g.Cout, _taps_r(g), g.S, g.Cin), dtype=torch.float32, device=dy.device)
    fn = lib().bdv_conv_wgrad_partial_pl if use_pl else lib().bdv_conv_wgrad_partial_x3 if use_x3 else lib().bdv_conv_wgrad_partial
    if use_pl:
        check(fn(_p(dy), _p(x), ctypes.byref(g), _p(slab), slab.numel() * 4, PIECES, _p(pre_bn[0] if pre_bn is not None else None),
                 _p(pre_bn[1] if pre_bn is not None else None), _stream()), 'bdv_conv_wgrad_partial_pl')
    else:
        check(fn(_p(dy), _p(x), ctypes.byref(g), _p(slab), slab.numel() * 4, _stream()), 'bdv_conv_wgrad_partial')
    if dw is None:
        dw = torch.empty((g.Cout, _taps_r(g), g.S, g.Cin), dtype=torch.float32, device=dy.device)
    return slab, dw


def wgrad_reduce_batched(items, beta: float = 0.0):
    """items: [(slab, dw), ...] from ``conv_wgrad_partial``; dw[k] = beta * dw[k] + sum over the slices of slab[k], for all
    items in as few launches as BDV_MAX_REDUCE_ITEMS allows."""
    MAXN = 32
    for a in range(0, len(items), MAXN):
        chunk = items[a:a + MAXN]
        n = len(chunk)
        for slab, dw in chunk:
            _chk(slab, name='slab')
            _chk(dw, name='dw')
            if slab.shape[1:] != dw.shape:
                raise ValueError(f'wgrad_reduce_batched: slab {tuple(slab.shape)} does not stack dw {tuple(dw.shape)}')
        slabs = (ctypes.c_void_p * n)(*[s.data_ptr() for s, _ in chunk])
        dws = (ctypes.c_void_p * n)(*[d.data_ptr() for _, d in chunk])
        splits = (ctypes.c_int * n)(*[s.shape[0] for s, _ in chunk])
        numels = (ctypes.c_int64 * n)(*[d.numel() for _, d in chunk])
        check(lib().bdv_wgrad_reduce_batched(ctypes.cast(slabs, ctypes.c_void_p), ctypes.cast(dws, ctypes.c_void_p),
                                             ctypes.cast(splits, ctypes.c_void_p), ctypes.cast(numels, ctypes.c_void_p), n,
                                             float(beta), _stream()), 'bdv_wgrad_reduce_batched')


# ---------------------------------------------------------------------------------------------
# heads / losses
# ---------------------------------------------------------------------------------------------

def lsc_fwd(x, w, K, P):
    _chk(x, name='x')
    N, D = x.shape
    _chk(w, (K, P * D), name='w')
    sim = torch.empty((N, K), dtype=torch.float32, device=x.device)
    xnorm = torch.empty(N, dtype=torch.float32, device=x.device)
    wnorm = torch.empty(K * P, dtype=torch.float32, device=x.device)
    cosbuf = torch.empty((N, K * P), dtype=torch.float32, device=x.device)
    check(lib().bdv_lsc_fwd(_p(x), _p(w), _p(sim), _p(xnorm), _p(wnorm), _p(cosbuf), N, D, K, P, _stream()), 'bdv_lsc_fwd')
    return sim, xnorm, wnorm, cosbuf


def lsc_bwd(dsim, x, w, xnorm, wnorm, cosbuf, K, P, need_dw=True, dw=None, beta_w=0.0):
    N, D = x.shape
    _chk(dsim, (N, K), name='dsim')
    _chk(x, name='x')
    _chk(w, (K, P * D), name='w')
    _chk(xnorm, (N,), name='xnorm')
    _chk(wnorm, (K * P,), name='wnorm')
    _chk(cosbuf, (N, K * P), name='cosbuf')
    dx = torch.empty_like(x)
    if need_dw and dw is None:
        dw = torch.empty_like(w)
        beta_w = 0.0
    if dw is not None:
        _chk(dw, (K, P * D), name='dw')
    ws = torch.empty((N, K * P), dtype=torch.float32, device=x.device)
    check(lib().bdv_lsc_bwd(_p(dsim), _p(x), _p(w), _p(xnorm), _p(wnorm), _p(cosbuf), _p(dx), _p(dw if need_dw else None),
                            float(beta_w), _p(ws), N, D, K, P, _stream()), 'bdv_lsc_bwd')
    return dx, (dw if need_dw else None)


def linear_fwd(x, w, b):
    _chk(x, name='x')
    N, D = x.shape
    K = w.shape[0]
    _chk(w, (K, D), name='w')
    if b is not None:
        _chk(b, (K,), name='b')
    out = torch.empty((N, K), dtype=torch.float32, device=x.device)
    check(lib().bdv_linear_fwd(_p(x), _p(w), _p(b), _p(out), N, D, K, _stream()), 'bdv_linear_fwd')
    return out


def linear_bwd(dout, x, w, need_dx=True, need_dw=True, need_db=True, dw=None, db=None, beta_w=0.0):
    N, D = x.shape
    K = w.shape[0]
    _chk(dout, (N, K), name='dout')
    _chk(x, name='x')
    _chk(w, (K, D), name='w')
    dx = torch.empty_like(x) if need_dx else None
    if need_dw and dw is None:
        dw = torch.empty_like(w)
        db = torch.empty(K, dtype=torch.float32, device=x.device) if need_db else None
        beta_w = 0.0
    if need_dw:
        _chk(dw, (K, D), name='dw')
        if db is not None:
            _chk(db, (K,), name='db')
    check(lib().bdv_linear_bwd(_p(dout), _p(x), _p(w), _p(dx), _p(dw if need_dw else None), _p(db if need_dw else None),
                               float(beta_w), N, D, K, _stream()), 'bdv_linear_bwd')
    return dx, (dw if need_dw else None), (db if need_dw else None)


def consensus_fwd(s, B, T):
    _chk(s, name='s')
    K = s.shape[1]
    if s.shape[0] != B * T:
        raise ValueError(f'consensus_fwd: {s.shape[0]} rows != B*T = {B * T}')
    out = torch.empty((B, K), dtype=torch.float32, device=s.device)
    check(lib().bdv_consensus_fwd(_p(s), _p(out), B, T, K, _stream()), 'bdv_consensus_fwd')
    return out


def consensus_bwd(dout, T):
    _chk(dout, name='dout')
    B, K = dout.shape
    ds = torch.empty((B * T, K), dtype=torch.float32, device=dout.device)
    check(lib().bdv_consensus_bwd(_p(dout), _p(ds), B, T, K, _stream()), 'bdv_consensus_bwd')
    return ds


def dropout(x, p, seed):
    _chk(x, name='x')
    out = torch.empty_like(x)
    check(lib().bdv_dropout(_p(x), _p(out), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()), 'bdv_dropout')
    return out


def lsc_loss(sim, targets, eta, margin, hinge, class_weights=None):
    _chk(sim, name='sim')
    B, K = sim.shape
    _chk(targets, (B,), dtype=torch.int64, name='targets')
    _chk(eta, (1,), name='eta')
    if class_weights is not None:
        _chk(class_weights, (K,), name='class_weights')
    out = torch.empty(2, dtype=torch.float32, device=sim.device)   # loss, deta
    dsim = torch.empty_like(sim)
    check(lib().bdv_lsc_loss(_p(sim), _p(targets), _p(eta), float(margin), int(bool(hinge)), _p(class_weights), _p(out[0:1]),
                             _p(dsim), _p(out[1:2]), B, K, _stream()), 'bdv_lsc_loss')
    return out[0], dsim, out[1:2]


def softce_loss(score, soft_targets=None, labels=None):
    _chk(score, name='score')
    B, K = score.shape
    if soft_targets is not None:
        _chk(soft_targets, (B, K), name='soft_targets')
    else:
        _chk(labels, (B,), dtype=torch.int64, name='labels')
    loss = torch.empty(1, dtype=torch.float32, device=score.device)
    dscore = torch.empty_like(score)
    check(lib().bdv_softce_loss(_p(score), _p(soft_targets), _p(labels), _p(loss), _p(dscore), B, K, _stream()),
          'bdv_softce_loss')
    return loss[0], dscore


def icarl_targets(labels, prev_logits, prev_K, K, base_targets=None):
    """(B, K) targets of ICARLModel.training_step: ``base_targets`` (or one-hot), old-class rows <- softmax(prev logits)."""
    B = labels.numel()
    labels = labels.reshape(B)
    _chk(labels, (B,), dtype=torch.int64, name='labels')
    if prev_logits is not None:
        _chk(prev_logits, (B, K), name='prev_logits')
    if base_targets is not None:
        _chk(base_targets, (B, K), name='base_targets')
    tgt = torch.empty((B, K), dtype=torch.float32, device=labels.device)
    check(lib().bdv_icarl_targets(_p(labels), _p(prev_logits), int(prev_K), _p(base_targets), _p(tgt), B, K, _stream()),
          'bdv_icarl_targets')
    return tgt


def acm_targets(labels, background_labels, foreground_ratio, alpha, K):
    """ACMSmoothCE smooth labels (libs/losses/acm_smooth_ce.py:18-28) -> (B, K)."""
    B = labels.numel()
    labels = labels.reshape(B)
    _chk(labels, (B,), dtype=torch.int64, name='labels')
    _chk(background_labels, (B,), dtype=torch.int64, name='background_labels')
    _chk(foreground_ratio, (B,), name='foreground_ratio')
    tgt = torch.empty((B, K), dtype=torch.float32, device=labels.device)
    check(lib().bdv_acm_targets(_p(labels), _p(background_labels), _p(foreground_ratio), float(alpha), _p(tgt), B, K, _stream()),
          'bdv_acm_targets')
    return tgt


def softmax_mean(s, B, n, apply_softmax=True):
    _chk(s, name='s')
    K = s.shape[1]
    if s.shape[0] != B * n:
        raise ValueError('softmax_mean: row count mismatch')
    out = torch.empty((B, K), dtype=torch.float32, device=s.device)
    check(lib().bdv_softmax_mean(_p(s), _p(out), B, n, K, int(bool(apply_softmax)), _stream()), 'bdv_softmax_mean')
    return out


def topk_acc(score, labels):
    _chk(score, name='score')
    B, K = score.shape
    _chk(labels, (B,), dtype=torch.int64, name='labels')
    acc = torch.empty(2, dtype=torch.float32, device=score.device)
    check(lib().bdv_topk_acc(_p(score), _p(labels), _p(acc), B, K, _stream()), 'bdv_topk_acc')
    return acc


def kd_mse_fwd(cur, prev):
    """cur/prev: any same-shape, same-stride dense tensors (compared in storage order)."""
    if cur.shape != prev.shape or cur.stride() != prev.stride() or cur.dtype != prev.dtype:
        raise ValueError('kd_mse: cur and prev need identical shape, strides and dtype')
    a, b = _dense_storage(cur), _dense_storage(prev)
    mse = torch.empty(1, dtype=torch.float32, device=cur.device)
    ws = workspace(lib().bdv_reduce_workspace_bytes(), cur.device, 'red')
    check(lib().bdv_kd_mse_fwd(_p(a), _p(b), _p(mse), a.numel(), _p(ws), ws.numel(), _act_code(a), _stream()), 'bdv_kd_mse_fwd')
    return mse[0]


def kd_mse_bwd(cur, prev, gscale_dev, gscale_host=1.0):
    a, b = _dense_storage(cur), _dense_storage(prev)
    d = torch.empty_like(cur)        # preserves strides for dense tensors
    if d.stride() != cur.stride():
        raise ValueError('kd_mse_bwd: could not preserve strides')
    if gscale_dev is not None:
        _chk(gscale_dev.reshape(1), (1,), name='gscale')
    check(lib().bdv_kd_mse_bwd(_p(a), _p(b), _p(gscale_dev), float(gscale_host), _p(_dense_storage(d)), a.numel(), _act_code(a), _stream()),
          'bdv_kd_mse_bwd')
    return d


def _dense_storage(t: torch.Tensor) -> torch.Tensor:
    """Flat view over the storage of a dense (possibly permuted) tensor."""
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('expected an fp32 (or, in the bf16-storage mode, bf16) GPU tensor; there is no CPU fallback')
    if t.is_contiguous():
        return t.reshape(-1)
    order = sorted(range(t.dim()), key=lambda i: -t.stride(i))
    p = t.permute(order)
    if not p.is_contiguous():
        raise ValueError('tensor is not dense')
    return p.reshape(-1)


# ---------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------

def multi_sqnorm(grad_ptrs, numels, ntensors, out):
    ws = workspace(ntensors * 32 * 4, out.device, 'sqn')
    check(lib().bdv_multi_sqnorm(_p(grad_ptrs), _p(numels), ntensors, _p(out), _p(ws), ws.numel(), _stream()), 'bdv_multi_sqnorm')


def clip_coef(sqnorm, grad_scale, max_norm, coef):
    check(lib().bdv_clip_coef(_p(sqnorm), float(grad_scale), float(max_norm), _p(coef), _stream()), 'bdv_clip_coef')


def multi_sgd(param_ptrs, grad_ptrs, buf_ptrs, numels, lrs, wds, ntensors, momentum, grad_scale, coef):
    check(lib().bdv_multi_sgd(_p(param_ptrs), _p(grad_ptrs), _p(buf_ptrs), _p(numels), _p(lrs), _p(wds), ntensors,
                              float(momentum), float(grad_scale), _p(coef), _stream()), 'bdv_multi_sgd')


# ---------------------------------------------------------------------------------------------
# representation path (clip representations, NME classifier, class means, herding)
# ---------------------------------------------------------------------------------------------

def repr_from_features(feat, B, crops, T):
    """feat (B*crops*T, D) -> (repr (B, crops, D), mean_crops (B, D)); libs/cil/cil.py:501-506,:564-571."""
    _chk(feat, name='feat')
    if feat.dim() != 2 or feat.shape[0] != B * crops * T:
        raise ValueError(f'repr_from_features: feat {tuple(feat.shape)} is not (B*crops*T = {B * crops * T}, D)')
    D = feat.shape[1]
    rp = torch.empty((B, crops, D), dtype=torch.float32, device=feat.device)
    mc = torch.empty((B, D), dtype=torch.float32, device=feat.device)
    check(lib().bdv_repr_from_features(_p(feat), _p(rp), _p(mc), B, crops, T, D, _stream()), 'bdv_repr_from_features')
    return rp, mc


def nme_classify(repr_, class_means):
    """repr_ (S, crops, D), class_means (K, D) -> (similarity (S, K), pred (S,) int64); libs/cil/cil.py:945-960."""
    _chk(repr_, name='repr_')
    _chk(class_means, name='class_means')
    if repr_.dim() != 3 or class_means.dim() != 2 or repr_.shape[2] != class_means.shape[1]:
        raise ValueError(f'nme_classify: repr_ {tuple(repr_.shape)} vs class_means {tuple(class_means.shape)}')
    S, crops, D = repr_.shape
    Kc = class_means.shape[0]
    sim = torch.empty((S, Kc), dtype=torch.float32, device=repr_.device)
    pred = torch.empty((S,), dtype=torch.int64, device=repr_.device)
    ws = workspace(lib().bdv_nme_workspace_bytes(Kc, D), repr_.device, 'nme')
    check(lib().bdv_nme_classify(_p(repr_), _p(class_means), _p(sim), _p(pred), S, crops, D, Kc, _p(ws), ws.numel(), _stream()),
          'bdv_nme_classify')
    return sim, pred


def class_means(repr_, labels, num_classes):
    """repr_ (n, D), labels (n,) int64 -> (num_classes, D) per-class means; libs/cil/cil.py:1079-1083."""
    _chk(repr_, name='repr_')
    _chk(labels, (repr_.shape[0],), dtype=torch.int64, name='labels')
    n, D = repr_.shape
    out = torch.empty((num_classes, D), dtype=torch.float32, device=repr_.device)
    check(lib().bdv_class_means(_p(repr_), _p(labels), _p(out), n, D, int(num_classes), _stream()), 'bdv_class_means')
    return out


def herding_select(features, num_exemplars, cosine_distance):
    """features (n, D) of one class -> (class_mean (1, D), indices (m,) int64, dist (m,)); memory_selection.py:70-92."""
    _chk(features, name='features')
    if features.dim() != 2:
        raise ValueError(f'herding_select: features must be (n, D), got {tuple(features.shape)}')
    n, D = features.shape
    m = int(num_exemplars)
    cm = torch.empty((1, D), dtype=torch.float32, device=features.device)
    idx = torch.empty((max(m, 1),), dtype=torch.int64, device=features.device)
    dist = torch.empty((max(m, 1),), dtype=torch.float32, device=features.device)
    ws = workspace(lib().bdv_herding_workspace_bytes(n, D), features.device, 'herding')
    check(lib().bdv_herding_select(_p(features), n, D, m, int(bool(cosine_distance)), _p(cm), _p(idx), _p(dist), _p(ws),
                                   ws.numel(), _stream()), 'bdv_herding_select')
    return cm, idx[:m], dist[:m]


# ---- data path: RandAugment on uint8 frames ---------------------------------------------------------------------------

RANDAUG_NUM_OPS = 12


def randaug_apply(frames_u8, op_i, op_d, out=None):
    """One operation slot of RandAugment for a batch of clips.  frames_u8 (B, T, H, W, 3) uint8; op_i (B, 8) int32 and
    op_d (B, 4) float64 device tables (layout: include/bdvcil_hip.h, bdv_randaug_apply) -> new (B, T, H, W, 3) uint8."""
    _chk(frames_u8, dtype=torch.uint8, name='frames_u8')
    if frames_u8.dim() != 5 or frames_u8.shape[-1] != 3:
        raise ValueError(f'randaug_apply: frames must be (B, T, H, W, 3), got {tuple(frames_u8.shape)}')
    B, T, H, W, _ = frames_u8.shape
    _chk(op_i, (B, 8), torch.int32, 'op_i')
    _chk(op_d, (B, 4), torch.float64, 'op_d')
    if out is None:
        out = torch.empty_like(frames_u8)
    else:
        _chk(out, frames_u8.shape, torch.uint8, 'out')
        if out.data_ptr() == frames_u8.data_ptr():
            raise ValueError('randaug_apply: out must not alias the input')
    ws = workspace(lib().bdv_randaug_workspace_bytes(B, T, H, W), frames_u8.device, 'randaug')
    check(lib().bdv_randaug_apply(_p(frames_u8), _p(out), _p(op_i), _p(op_d), B, T, H, W, _p(ws), ws.numel(), _stream()),
          'bdv_randaug_apply')
    return out
