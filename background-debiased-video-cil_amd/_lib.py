"""ctypes binding of ``csrc/libbdvcil_hip.so`` (C ABI declared in ``include/bdvcil_hip.h``).

The library is loaded lazily and per process (a ctypes handle cannot be pickled; modules that are
shipped to ``ddp_spawn`` workers re-open it on first use -- SURVEY.md section 8(b) "Threading").
There is NO fallback: if the shared object is missing or a call fails, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint16, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# BDVCIL_LIB_PATH: load another build of the library (A/B runs of two kernel versions in one gpurun call); the source-hash check
# is then the caller's business
_OVERRIDE = os.environ.get('BDVCIL_LIB_PATH')
LIB_PATH = _OVERRIDE or os.path.join(_HERE, 'csrc', 'libbdvcil_hip.so')
ABI_VERSION = 29

_lib = None


class ConvGeom(Structure):
    """Mirror of ``bdv_conv_geom``."""
    _fields_ = [(n, c_int32) for n in
                ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'R', 'S', 'stride', 'pad', 'T', 'fold', 'pad_w', 'Rt', 'st_t', 'act_dtype')]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


P = c_void_p
_F3 = c_float * 3


class ConvAffine(Structure):
    """Mirror of ``bdv_conv_affine``."""
    _fields_ = [('scale', c_void_p), ('shift', c_void_p), ('residual', c_void_p), ('relu', c_int32)]


class BnStatFuse(Structure):
    """Mirror of ``bdv_bn_stat_fuse``."""
    _fields_ = [('y', c_void_p), ('relu_mask', c_void_p), ('mean', c_void_p), ('invstd', c_void_p), ('partial', c_void_p),
                ('relu_scale', c_void_p), ('relu_shift', c_void_p)]

class JpegInfo(Structure):
    """Mirror of ``bdv_jpeg_info``."""
    _fields_ = [('width', c_int32), ('height', c_int32), ('ncomp', c_int32), ('h', c_int32 * 3), ('v', c_int32 * 3),
                ('blocks_w', c_int32 * 3), ('blocks_h', c_int32 * 3), ('down_w', c_int32 * 3), ('down_h', c_int32 * 3),
                ('qt', (c_uint16 * 64) * 3), ('coef_offset', c_int64 * 3), ('coef_count', c_int64)]

    def geometry_key(self):
        """Everything ``bdv_jpeg_reconstruct_u8`` needs equal across a batch (all but the quantisation tables)."""
        return (self.width, self.height, self.ncomp, tuple(self.h), tuple(self.v))


# name -> (restype, argtypes)
SIGNATURES = {
    'bdv_last_error': (c_char_p, []),
    'bdv_abi_version': (c_int, []),
    'bdv_source_hash': (c_char_p, []),
    'bdv_conv_workspace_bytes': (c_size_t, [POINTER(ConvGeom), c_int]),
    'bdv_conv_fprop_stat_rows': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_fprop': (c_int, [P, P, P, POINTER(ConvGeom), P, POINTER(ConvAffine), P, c_size_t, P]),
    'bdv_conv_fprop_x3': (c_int, [P, P, P, POINTER(ConvGeom), P, POINTER(ConvAffine), P, c_size_t, P]),
    'bdv_conv_dgrad_stat_rows': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_dgrad': (c_int, [P, P, P, P, P, POINTER(ConvGeom), POINTER(BnStatFuse), P, c_size_t, P]),
    'bdv_conv_dgrad_x3': (c_int, [P, P, P, P, P, P, POINTER(ConvGeom), POINTER(BnStatFuse), P, c_size_t, P]),
    'bdv_conv_weight_planes_bytes': (c_size_t, [POINTER(ConvGeom)]),
    'bdv_conv_split_weights': (c_int, [P, POINTER(ConvGeom), P, P, P]),
    'bdv_conv_fprop_pre_ok': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_fprop_pre_stat_rows': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_wgrad_pre_ok': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_debug_force_tile': (c_int, [c_int]),
    'bdv_conv_uses_planes': (c_int, [POINTER(ConvGeom), c_int, c_int]),
    'bdv_conv_kernel_name': (c_int, [POINTER(ConvGeom), c_int, c_int, c_char_p, c_size_t]),
    'bdv_conv_fprop_pl_stat_rows': (c_int, [POINTER(ConvGeom), c_int]),
    'bdv_conv_dgrad_pl_stat_rows': (c_int, [POINTER(ConvGeom), c_int]),
    'bdv_conv_fprop_pl': (c_int, [P, P, P, P, POINTER(ConvGeom), P, POINTER(ConvAffine), P, c_size_t, c_int, P, P, P]),
    'bdv_conv_dgrad_pl': (c_int, [P, P, P, P, P, P, POINTER(ConvGeom), POINTER(BnStatFuse), P, c_size_t, c_int, P]),
    'bdv_conv_wgrad': (c_int, [P, P, P, c_float, POINTER(ConvGeom), P, c_size_t, P]),
    'bdv_conv_wgrad_splits': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_wgrad_partial': (c_int, [P, P, POINTER(ConvGeom), P, c_size_t, P]),
    'bdv_conv_wgrad_partial_x3': (c_int, [P, P, POINTER(ConvGeom), P, c_size_t, P]),
    'bdv_conv_wgrad_pl_splits': (c_int, [POINTER(ConvGeom)]),
    'bdv_conv_wgrad_partial_pl': (c_int, [P, P, POINTER(ConvGeom), P, c_size_t, c_int, P, P, P]),
    'bdv_wgrad_reduce_batched': (c_int, [P, P, P, P, c_int, c_float, P]),
    'bdv_bn_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'bdv_bn_train_stats': (c_int, [P, c_int64, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, c_size_t, P]),
    'bdv_bn_train_finalize': (c_int, [P, c_int, c_int64, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P]),
    'bdv_bn_eval_params': (c_int, [c_int, P, P, P, P, c_float, P, P, P]),
    'bdv_bn_apply': (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    'bdv_bn_backward': (c_int, [P, P, P, P, P, P, P, P, P, c_float, c_int64, c_int, c_int, P, c_int, P, P, P, c_size_t, c_int, P]),
    'bdv_bn_backward_maxpool': (c_int, [P, P, P, P, P, P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    'bdv_relu_bwd': (c_int, [P, P, P, P, c_int64, c_int, P]),
    'bdv_add': (c_int, [P, P, P, c_int64, c_int, P]),
    'bdv_nchw3_to_nhwc4': (c_int, [P, P, c_int, c_int, c_int, P]),
    'bdv_maxpool_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'bdv_maxpool_bwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'bdv_maxpool_t2_fwd': (c_int, [P, P, P, c_int64, c_int64, P]),
    'bdv_maxpool_t2_bwd': (c_int, [P, P, P, c_int64, c_int64, P]),
    'bdv_bn_relu_maxpool_fwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'bdv_avgpool_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_avgpool_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_bgmix_normalize_u8': (c_int, [P, P, c_int, P, c_float, _F3, _F3, _F3, P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_bg_resize_crop_u8': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P, P]),
    'bdv_resize_linear_u8': (c_int, [P, c_int, c_int, c_int, P, c_int, P, P, c_int, c_int, P]),
    'bdv_jpeg_parse': (c_int, [P, c_size_t, POINTER(JpegInfo)]),
    'bdv_jpeg_entropy_decode': (c_int, [P, c_size_t, POINTER(JpegInfo), P]),
    'bdv_jpeg_entropy_decode_batch': (c_int, [P, P, c_int, POINTER(JpegInfo), P, P, c_int]),
    'bdv_jpeg_workspace_bytes': (c_size_t, [POINTER(JpegInfo), c_int]),
    'bdv_jpeg_reconstruct_u8': (c_int, [P, P, POINTER(JpegInfo), c_int, P, c_size_t, P, P]),
    'bdv_crop_normalize_u8': (c_int, [P, P, c_int, c_int, c_int, _F3, _F3, P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_lsc_fwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_lsc_bwd': (c_int, [P, P, P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, c_int, P]),
    'bdv_linear_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'bdv_linear_bwd': (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, c_int, P]),
    'bdv_consensus_fwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'bdv_consensus_bwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'bdv_dropout': (c_int, [P, P, c_int64, c_float, c_uint64, P]),
    'bdv_lsc_loss': (c_int, [P, P, P, c_float, c_int, P, P, P, P, c_int, c_int, P]),
    'bdv_softce_loss': (c_int, [P, P, P, P, P, c_int, c_int, P]),
    'bdv_icarl_targets': (c_int, [P, P, c_int, P, P, c_int, c_int, P]),
    'bdv_acm_targets': (c_int, [P, P, P, c_float, P, c_int, c_int, P]),
    'bdv_softmax_mean': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_topk_acc': (c_int, [P, P, P, c_int, c_int, P]),
    'bdv_repr_from_features': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'bdv_nme_workspace_bytes': (c_size_t, [c_int, c_int]),
    'bdv_nme_classify': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'bdv_class_means': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'bdv_herding_workspace_bytes': (c_size_t, [c_int, c_int]),
    'bdv_herding_select': (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    'bdv_randaug_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'bdv_randaug_apply': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'bdv_reduce_workspace_bytes': (c_size_t, []),
    'bdv_kd_mse_fwd': (c_int, [P, P, P, c_int64, P, c_size_t, c_int, P]),
    'bdv_kd_mse_bwd': (c_int, [P, P, P, c_float, P, c_int64, c_int, P]),
    'bdv_multi_sqnorm': (c_int, [P, P, c_int, P, P, c_size_t, P]),
    'bdv_clip_coef': (c_int, [P, c_float, c_float, P, P]),
    'bdv_multi_sgd': (c_int, [P, P, P, P, P, P, c_int, c_float, c_float, P, P]),
}


class HipExtensionError(RuntimeError):
    pass


HASHED_SOURCES = ('conv_mfma.hip', 'bn.hip', 'pool_frontend.hip', 'head_loss.hip', 'repr.hip', 'augment.hip', 'optim.hip', 'jpeg.hip',
                  'api_common.cpp', 'common.h', 'Makefile', '../../include/bdvcil_hip.h')   # = HASHED in csrc/Makefile


def source_hash():
    """sha256 of the in-tree sources in the Makefile's order (None when a binary-only install carries no sources)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(_HERE, 'csrc')
    for name in HASHED_SOURCES:
        path = os.path.join(base, name)
        if not os.path.exists(path):
            return None
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def lib():
    """Return the loaded library, loading it on first use.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f'HIP extension not built: {LIB_PATH} is missing. Run `python -c "import __graft_entry__ as g; '
                f'g.build()"` (or `make -C {os.path.dirname(LIB_PATH)}`). There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.bdv_abi_version() != ABI_VERSION:
            raise HipExtensionError(f'ABI mismatch: library {handle.bdv_abi_version()} != binding {ABI_VERSION}')
        want, have = source_hash(), handle.bdv_source_hash().decode()
        if want is not None and want != have and not _OVERRIDE:
            raise HipExtensionError(f'stale library: {LIB_PATH} was built from other sources (hash {have[:12]}, in-tree '
                                    f'{want[:12]}). Rebuild: make -C {os.path.dirname(LIB_PATH)}')
        _lib = handle
    return _lib


def check(code: int, what: str = ''):
    if code != 0:
        msg = lib().bdv_last_error()
        raise HipExtensionError(f'{what or "bdvcil_hip"} failed with code {code}: {msg.decode() if msg else ""}')
