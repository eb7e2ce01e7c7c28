"""``RawFrameDecode`` of the configs' pipelines on the GPU: JPEG bytes -> uint8 RGB frames in HBM.

The reference decodes on CPU workers (UPSTREAM mmaction2 ``RawFrameDecode`` -> ``mmcv.imfrombytes(channel_order='rgb')`` ->
``cv2.imdecode``: libjpeg-turbo's defaults; configs/ucf101/bgmix_plus_randAug/...py:126).  Here the Huffman stage -- a serial bit
stream -- runs on host threads (``bdv_jpeg_entropy_decode``; ctypes releases the GIL, one image per thread), the coefficients cross
PCIe once as int16 (about the size of the decoded image), and dequantisation, inverse DCT, chroma upsampling and colour conversion
run as two launches for the whole batch (``bdv_jpeg_reconstruct_u8``), bit-identical to libjpeg-turbo (tests/test_jpeg_gpu.py).
The output is the ``(N, H, W, 3)`` uint8 layout ``RandAugment`` / ``TrainClipFrontEnd`` / ``CropFrontEnd`` take."""
from __future__ import annotations

import ctypes
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np
import torch

from ._lib import JpegInfo, check, lib


def jpeg_parse(data: bytes) -> JpegInfo:
    """Header of a JPEG stream (sizes, sampling factors, block grids, quantisation tables).  Raises on anything the decoder does
    not cover (progressive / arithmetic-coded / CMYK / 12-bit streams), with the reason."""
    info = JpegInfo()
    check(lib().bdv_jpeg_parse(data, len(data), ctypes.byref(info)), 'bdv_jpeg_parse')
    return info


def jpeg_entropy_decode(data: bytes, info: JpegInfo = None, out: np.ndarray = None) -> Tuple[JpegInfo, np.ndarray]:
    """Host stage: the stream's quantised DCT coefficients, ``info.coef_count`` int16 values (component rasters of 64-value blocks
    in natural order).  ``out``: a C-contiguous int16 buffer to fill (a row of the batch's pinned staging tensor)."""
    if info is None:
        info = jpeg_parse(data)
    if out is None:
        out = np.empty(info.coef_count, dtype=np.int16)
    assert out.dtype == np.int16 and out.flags['C_CONTIGUOUS'] and out.size == info.coef_count
    check(lib().bdv_jpeg_entropy_decode(data, len(data), ctypes.byref(info), out.ctypes.data), 'bdv_jpeg_entropy_decode')
    return info, out


class JpegDecoder:
    """Batched decode of equal-geometry JPEG streams (the frames of a rawframe dataset share one size and sampling).

    ``decode(streams) -> (N, H, W, 3) uint8`` on ``device``; ``decode_clips(clips) -> (B, T, H, W, 3)``.  Streams of different
    geometry in one call are decoded group by group and must then share ``(H, W)`` (else ``ValueError``: a batch tensor needs one
    size -- resize first, as the reference's pipeline does per sample)."""

    def __init__(self, device='cuda', threads: int = 8):
        self.device = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(threads)))

    def _group(self, streams: Sequence[bytes]):
        infos = list(self.pool.map(jpeg_parse, streams))
        groups = {}
        for i, inf in enumerate(infos):
            groups.setdefault(inf.geometry_key(), []).append(i)
        return infos, groups

    def decode(self, streams: Sequence[bytes]) -> torch.Tensor:
        if len(streams) == 0:
            raise ValueError('JpegDecoder.decode: empty batch')
        infos, groups = self._group(streams)
        sizes = {(k[0], k[1]) for k in groups}
        if len(sizes) != 1:
            raise ValueError(f'JpegDecoder.decode: images of different sizes in one batch: {sorted(sizes)}')
        W, H = next(iter(sizes))
        out = torch.empty(len(streams), H, W, 3, dtype=torch.uint8, device=self.device)
        for idx in groups.values():
            rgb = self._decode_group([streams[i] for i in idx], [infos[i] for i in idx])
            if len(groups) == 1:
                return rgb
            out[torch.as_tensor(idx, device=self.device)] = rgb
        return out

    def _decode_group(self, streams: List[bytes], infos: List[JpegInfo]) -> torch.Tensor:
        n, info = len(streams), infos[0]
        pin = self.device.type == 'cuda'
        coefs = torch.empty(n, info.coef_count, dtype=torch.int16, pin_memory=pin)
        qts = torch.zeros(n, 3, 64, dtype=torch.int16, pin_memory=pin)
        cn, qn = coefs.numpy(), qts.numpy().view(np.uint16)

        def work(i):
            jpeg_entropy_decode(streams[i], infos[i], cn[i])
            qn[i] = np.ctypeslib.as_array(infos[i].qt)
        list(self.pool.map(work, range(n)))
        coefs_d, qts_d = coefs.to(self.device, non_blocking=True), qts.to(self.device, non_blocking=True)
        ws_bytes = lib().bdv_jpeg_workspace_bytes(ctypes.byref(info), n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        rgb = torch.empty(n, info.height, info.width, 3, dtype=torch.uint8, device=self.device)
        from .kernels import _p, _stream
        check(lib().bdv_jpeg_reconstruct_u8(_p(coefs_d), _p(qts_d), ctypes.byref(info), n, _p(ws), ws_bytes, _p(rgb), _stream()),
              'bdv_jpeg_reconstruct_u8')
        return rgb      # (torch's pinned-memory allocator keeps the staging blocks until the asynchronous copies have run)

    def decode_clips(self, clips: Sequence[Sequence[bytes]]) -> torch.Tensor:
        """``clips``: B lists of T streams each -> (B, T, H, W, 3) uint8."""
        T = len(clips[0])
        if any(len(c) != T for c in clips):
            raise ValueError('JpegDecoder.decode_clips: clips of different lengths')
        flat = [s for c in clips for s in c]
        rgb = self.decode(flat)
        return rgb.view(len(clips), T, *rgb.shape[1:])
