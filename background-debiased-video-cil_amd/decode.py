"""``RawFrameDecode`` of the configs' pipelines on the GPU: JPEG bytes -> uint8 RGB frames in HBM.

The reference decodes on CPU workers (UPSTREAM mmaction2 ``RawFrameDecode`` -> ``mmcv.imfrombytes(channel_order='rgb')`` ->
``cv2.imdecode``: libjpeg-turbo's defaults; configs/ucf101/bgmix_plus_randAug/...py:126).  Here the Huffman stage -- a serial bit
stream -- runs on host threads (``bdv_jpeg_entropy_decode_batch``: std::thread workers inside the library, one image each), the coefficients cross
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
        self.threads = max(1, int(threads))
        self.pool = ThreadPoolExecutor(max_workers=self.threads)      # file reads and header parses; the Huffman stage threads in C

    def _staging(self, ncoef: int, nqt: int):
        """Two pinned staging sets used in turn (a fresh pinned allocation per batch costs several milliseconds, and torch's
        host allocator cannot hand a block back while its upload is still queued): (coefs, qts, event) -- the event marks the
        last upload from the set; waited on before the host threads overwrite it."""
        pin = self.device.type == 'cuda'
        if not hasattr(self, '_sets'):
            self._sets, self._turn = [None, None], 0
        self._turn ^= 1
        cur = self._sets[self._turn]
        if cur is None or cur[0].numel() < ncoef or cur[1].numel() < nqt:
            cur = (torch.empty(max(ncoef, cur[0].numel() if cur else 0), dtype=torch.int16, pin_memory=pin),
                   torch.empty(max(nqt, cur[1].numel() if cur else 0), dtype=torch.int16, pin_memory=pin),
                   torch.cuda.Event() if pin else None)
            self._sets[self._turn] = cur
        elif cur[2] is not None:
            cur[2].synchronize()
        return cur

    def _group(self, streams: Sequence[bytes]):
        infos = list(self.pool.map(jpeg_parse, streams))
        groups = {}
        for i, inf in enumerate(infos):
            groups.setdefault(inf.geometry_key(), []).append(i)
        return infos, groups

    def decode(self, streams: Sequence[bytes]) -> torch.Tensor:
        if len(streams) == 0:
            raise ValueError('JpegDecoder.decode: empty batch')
        # the common case -- every stream has the first one's geometry (frames of one dataset) -- needs no per-stream Python work: the
        # batch call checks each stream against `info` itself and names the first that differs; only then are the streams grouped
        try:
            return self._decode_group(list(streams), [jpeg_parse(streams[0])])
        except RuntimeError as e:
            if 'does not have the geometry' not in str(e):
                raise
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
        coefs, qts, done = self._staging(n * info.coef_count, n * 192)
        coefs, qts = coefs[:n * info.coef_count].view(n, info.coef_count), qts[:n * 192].view(n, 3, 64)
        ptrs = (ctypes.c_char_p * n)(*streams)
        sizes = (ctypes.c_size_t * n)(*[len(s) for s in streams])
        check(lib().bdv_jpeg_entropy_decode_batch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), n, ctypes.byref(info),
                                                  coefs.data_ptr(), qts.data_ptr(), self.threads), 'bdv_jpeg_entropy_decode_batch')
        coefs_d, qts_d = coefs.to(self.device, non_blocking=True), qts.to(self.device, non_blocking=True)
        ws_bytes = lib().bdv_jpeg_workspace_bytes(ctypes.byref(info), n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        rgb = torch.empty(n, info.height, info.width, 3, dtype=torch.uint8, device=self.device)
        from .kernels import _p, _stream
        check(lib().bdv_jpeg_reconstruct_u8(_p(coefs_d), _p(qts_d), ctypes.byref(info), n, _p(ws), ws_bytes, _p(rgb), _stream()),
              'bdv_jpeg_reconstruct_u8')
        if done is not None:
            done.record()         # the staging set may be refilled once the stream has passed the uploads
        return rgb

    def decode_clips(self, clips: Sequence[Sequence[bytes]]) -> torch.Tensor:
        """``clips``: B lists of T streams each -> (B, T, H, W, 3) uint8."""
        T = len(clips[0])
        if any(len(c) != T for c in clips):
            raise ValueError('JpegDecoder.decode_clips: clips of different lengths')
        flat = [s for c in clips for s in c]
        rgb = self.decode(flat)
        return rgb.view(len(clips), T, *rgb.shape[1:])


def rescale_size(w: int, h: int, scale) -> Tuple[int, int]:
    """UPSTREAM ``mmcv.rescale_size((w, h), scale)``: ``(-1, S)`` = short edge to S; sizes ``int(x * factor + 0.5)``."""
    a, b = scale
    long_edge, short_edge = (float('inf'), max(a, b)) if (a == -1 or b == -1) else (max(a, b), min(a, b))
    factor = min(long_edge / max(h, w), short_edge / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def sample_frames(total_frames: int, num_clips: int = 8, clip_len: int = 1, frame_interval: int = 1, test_mode: bool = False,
                  start_index: int = 1) -> np.ndarray:
    """UPSTREAM mmaction2 0.24 ``SampleFrames`` as the configs use it (``clip_len=1, frame_interval=1, num_clips=8``, no temporal
    jitter, ``out_of_bound_opt='loop'``): 1-based numbers of the ``img_{:05}.jpg`` files; the train form draws from ``np.random``."""
    ori = clip_len * frame_interval
    if test_mode:
        avg = (total_frames - ori + 1) / float(num_clips)
        offsets = (np.arange(num_clips) * avg + avg / 2.0).astype(np.int64) if total_frames > ori - 1 else np.zeros((num_clips,), dtype=np.int64)
    else:
        avg = (total_frames - ori + 1) // num_clips
        if avg > 0:
            offsets = np.arange(num_clips) * avg + np.random.randint(avg, size=num_clips)
        elif total_frames > max(num_clips, ori):
            offsets = np.sort(np.random.randint(total_frames - ori + 1, size=num_clips))
        elif avg == 0:
            offsets = np.around(np.arange(num_clips) * ((total_frames - ori + 1.0) / num_clips))
        else:
            offsets = np.zeros((num_clips,), dtype=np.int64)
    inds = (offsets[:, None] + np.arange(clip_len)[None, :] * frame_interval).reshape(-1)
    return (np.mod(inds, total_frames).astype(np.int64) + start_index)


class RawFrameClipLoader:
    """The configs' four pipelines (configs/ucf101/bgmix_plus_randAug/bgmix_seed_1000_inc_10_stages_bgmix_plus_randAug.py:124-182) for a
    whole batch, as the ``clip_loader`` of ``CILTaskLoop``: ``video_infos`` + phase -> the collated batch dict, frames never leaving HBM
    after the coefficient upload.

      train:                SampleFrames -> RawFrameDecode -> Resize(-1, 256) -> RandAugment -> MultiScaleCrop + Resize(224) -> Normalize,
                            then the background mix of BackgroundMixDataset for the samples RandAugment skipped (comix_loader.py:105-145)
      val / features_extraction: SampleFrames(test_mode) -> decode -> Resize(-1, 256) -> CenterCrop(224) -> Normalize
      test:                 SampleFrames(test_mode) -> decode -> Resize(-1, 256) -> TenCrop(256) -> Normalize

    Host work per batch: reading the files, the Huffman stage (thread pool) and the random draws -- per sample in the reference's order
    (frame offsets from ``np.random``; RandAugment's own draws; crop size and offset from ``random``; background index and crop from
    torch's generator), though a multi-worker DataLoader interleaves samples differently anyway.  ``bg_files``: JPEG backgrounds
    (``back_ground_from_bg_dir``); without it a random frame of a random video of the batch's dataset serves (comix_loader.py:134-137
    draws it from the whole dataset: pass ``bg_video_infos``)."""

    def __init__(self, device='cuda', filename_tmpl: str = 'img_{:05}.jpg', num_segments: int = 8, start_index: int = 1,
                 short_edge: int = 256, input_size: int = 224, randAug=None, randAug_prob: float = 0.75, alpha: float = 0.5,
                 multi_scale_crop: dict = None, bg_files: Sequence[str] = None, bg_video_infos: Sequence[dict] = None,
                 bg_resize: int = 256, test_crop=('TenCrop', 256), threads: int = 8):
        from .augment import RandAugment
        from .frontend import BackgroundCropFrontEnd, BackgroundMixFrontEnd, CropFrontEnd, MultiScaleCropResize, TrainClipFrontEnd
        self.device = torch.device(device)
        self.tmpl, self.T, self.start_index, self.short_edge = filename_tmpl, int(num_segments), int(start_index), int(short_edge)
        self.decoder = JpegDecoder(self.device, threads)
        msc = dict(input_size=input_size, scales=(1, 0.875, 0.75, 0.66), random_crop=False, max_wh_scale_gap=1, num_fixed_crops=13)
        msc.update(multi_scale_crop or {})
        self.train_front = TrainClipFrontEnd(randAug if randAug is not None else RandAugment(2, 10, randAug_prob), alpha=alpha,
                                             with_randAug=True, crop_resize=MultiScaleCropResize(**msc))
        self.bg_front = BackgroundCropFrontEnd(bg_resize, (input_size, input_size))
        self.center = CropFrontEnd('CenterCrop', input_size)
        self.test = CropFrontEnd(*test_crop)
        self.bg_files, self.bg_video_infos = (list(bg_files) if bg_files else None), bg_video_infos
        self.input_size = int(input_size)

    # ---- host side ------------------------------------------------------------------------------------------------------------
    def _read(self, path: str) -> bytes:
        with open(path, 'rb') as f:
            return f.read()

    def _frames(self, video_infos: List[dict], test_mode: bool):
        import os.path as osp
        inds = [sample_frames(int(v['total_frames']), self.T, test_mode=test_mode, start_index=self.start_index) for v in video_infos]
        paths = [osp.join(v['frame_dir'], self.tmpl.format(int(i))) for v, ii in zip(video_infos, inds) for i in ii]
        streams = list(self.decoder.pool.map(self._read, paths))
        clips = [streams[k * self.T:(k + 1) * self.T] for k in range(len(video_infos))]
        return self.decoder.decode_clips(clips), np.stack(inds)

    def _backgrounds(self, B: int, video_infos: List[dict]) -> torch.Tensor:
        """One background per sample, cropped: (B, S, S, 3) fp32 pixel values (only the mixed samples' rows are read)."""
        import os.path as osp
        import random
        paths = []
        for _ in range(B):
            if self.bg_files:
                paths.append(self.bg_files[int(torch.randint(len(self.bg_files), (1,)).item())])
            else:
                v = random.choice(self.bg_video_infos or video_infos)
                idx = random.randint(self.start_index, int(v['total_frames']) - 1 + self.start_index)
                paths.append(osp.join(v['frame_dir'], self.tmpl.format(idx)))
        streams = list(self.decoder.pool.map(self._read, paths))
        infos = [jpeg_parse(s) for s in streams]
        out = torch.empty(B, self.input_size, self.input_size, 3, dtype=torch.float32, device=self.device)
        groups = {}
        for i, inf in enumerate(infos):
            groups.setdefault((inf.width, inf.height), []).append(i)
        for idx in groups.values():            # the crop draws stay in sample order inside a size group; sizes rarely differ
            bg = self.decoder.decode([streams[i] for i in idx])
            out[torch.as_tensor(idx, device=self.device)] = self.bg_front(bg)
        return out

    def __call__(self, video_infos: List[dict], phase: str):
        B = len(video_infos)
        frames, inds = self._frames(video_infos, test_mode=phase != 'train')
        H, W = int(frames.shape[2]), int(frames.shape[3])
        Wr, Hr = rescale_size(W, H, (-1, self.short_edge))
        from . import kernels as K
        frames = K.resize_linear_u8(frames, Hr, Wr)
        extra = {}
        if phase == 'train':
            bg = self._backgrounds(B, video_infos)
            imgs, rand_flags, _ = self.train_front(frames, bg, as_nchw=True)
            extra['randAug'] = rand_flags
        elif phase == 'test':
            imgs = self.test.as_nchw(frames)
        else:
            imgs = self.center.as_nchw(frames)
        dev = self.device
        return {**extra, 'imgs': imgs,
                'label': torch.tensor([[v['label']] for v in video_infos], dtype=torch.int64, device=dev),
                'frame_dir': [v['frame_dir'] for v in video_infos],
                'total_frames': torch.tensor([int(v['total_frames']) for v in video_infos], dtype=torch.int64, device=dev),
                'clip_len': torch.ones(B, dtype=torch.int64, device=dev),
                'num_clips': torch.full((B,), self.T, dtype=torch.int64, device=dev),
                'frame_inds': torch.from_numpy(inds).to(dev)}


class PrefetchLoader:
    """Runs a ``clip_loader`` one batch ahead on a worker thread and its own HIP stream, so that reading, the Huffman stage, the
    per-sample draws and the decode / augment kernels of batch i + 1 overlap the training step of batch i (the reference gets the same
    from its DataLoader workers).  ``submit(video_infos, phase)`` queues a batch, ``get()`` returns the oldest queued batch after making
    the caller's stream wait for it; or iterate: ``for batch in PrefetchLoader(loader).iterate(list_of_video_info_lists, phase)``.
    The batch's tensors are handed to the consumer's stream with ``record_stream`` (a handful of tensors per step)."""

    def __init__(self, loader, depth: int = 2):
        self.loader, self.depth = loader, max(1, int(depth))
        self.stream = torch.cuda.Stream()
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.queue = []

    def _work(self, video_infos, phase):
        with torch.cuda.stream(self.stream):
            batch = self.loader(video_infos, phase)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return batch, ev

    def submit(self, video_infos, phase: str):
        self.queue.append(self.pool.submit(self._work, video_infos, phase))

    def get(self):
        batch, ev = self.queue.pop(0).result()
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return batch

    def iterate(self, batches_of_infos, phase: str):
        it = iter(batches_of_infos)
        for infos in it:
            self.submit(infos, phase)
            if len(self.queue) >= self.depth:
                break
        for infos in it:
            out = self.get()
            self.submit(infos, phase)
            yield out
        while self.queue:
            yield self.get()
