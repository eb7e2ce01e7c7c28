"""Throughput of the JPEG decode stage (bdvcil_amd.decode) on one box: host Huffman stage per thread count, the two device kernels
(HIP events) against their HBM traffic, and the whole decode of a 32 x 8 clip batch of 240 x 320 4:2:0 frames.  Dev tool, GPU only.
    python tools/bench_jpeg.py > gpurun_out/jpeg_decode.txt"""
import ctypes
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from bdvcil_amd import kernels as K
from bdvcil_amd._lib import check, lib
from bdvcil_amd.decode import JpegDecoder, jpeg_entropy_decode, jpeg_parse

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:240, 0:320]
streams = []
for i in range(256):
    a = np.stack([128 + 100 * np.sin(xx / (7.0 + i % 5) + yy / 13.0), 128 + 90 * np.cos(xx / 5.0 + i), 128 + 80 * np.sin(yy / 3.0 + xx / 11.0)], -1)
    a = np.clip(a + rng.normal(0, 10, a.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, 'JPEG', quality=85, subsampling=2)
    streams.append(buf.getvalue())
nbytes = sum(len(s) for s in streams)
print(f'256 frames of 240 x 320, 4:2:0, quality 85: {nbytes / 256 / 1024:.1f} KB per stream')

info = jpeg_parse(streams[0])
t0 = time.perf_counter()
for s in streams:
    jpeg_entropy_decode(s)
dt = time.perf_counter() - t0
print(f'host Huffman stage, one thread: {256 / dt:.0f} frames/s ({dt / 256 * 1e3:.2f} ms per frame, {nbytes / dt / 1e6:.1f} MB/s of stream)')
t0 = time.perf_counter()
for s in streams:
    np.asarray(Image.open(io.BytesIO(s)).convert('RGB'))
dt = time.perf_counter() - t0
print(f'Pillow / libjpeg-turbo full decode, one thread (the reference\'s CPU path per worker): {256 / dt:.0f} frames/s')
for threads in (1, 2, 4, 8, 16):
    dec = JpegDecoder(dev, threads)
    dec.decode(streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = dec.decode(streams)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'JpegDecoder.decode, {threads:2d} host threads: {256 / dt:.0f} frames/s = {32 / dt:.0f} clips/s of 8 frames ({dt * 1e3:.1f} ms per 32 x 8 batch)')

# device stage alone
n = 256
coefs = torch.empty(n, info.coef_count, dtype=torch.int16)
qts = torch.zeros(n, 3, 64, dtype=torch.int16)
for i, s in enumerate(streams):
    inf, _ = jpeg_entropy_decode(s, None, coefs.numpy()[i])
    qts.numpy().view(np.uint16)[i] = np.ctypeslib.as_array(inf.qt)
cd, qd = coefs.to(dev), qts.to(dev)
ws_bytes = lib().bdv_jpeg_workspace_bytes(ctypes.byref(info), n)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
rgb = torch.empty(n, 240, 320, 3, dtype=torch.uint8, device=dev)
run = lambda: check(lib().bdv_jpeg_reconstruct_u8(K._p(cd), K._p(qd), ctypes.byref(info), n, K._p(ws), ws_bytes, K._p(rgb), K._stream()), 'rec')
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
traffic = cd.numel() * 2 + 2 * ws_bytes + rgb.numel()          # coefficients in, planes out and in again, RGB out
print(f'device stage (inverse DCT + upsampling + colour, two launches), 256 frames: {ms * 1e3:.0f} us = {256 / ms * 1e3:.0f} frames/s, '
      f'{traffic / ms / 1e6:.0f} GB/s over {traffic / 1e6:.1f} MB of algorithmic traffic')
frames = rgb.view(32, 8, 240, 320, 3)
for _ in range(3):
    K.resize_linear_u8(frames, 256, 341)
e0.record()
for _ in range(20):
    r = K.resize_linear_u8(frames, 256, 341)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'Resize(-1, 256) of the batch (240 x 320 -> 256 x 341): {ms * 1e3:.0f} us, {(frames.numel() + r.numel()) / ms / 1e6:.0f} GB/s')

# the whole train pipeline of the configs from files: SampleFrames -> read -> decode -> Resize(-1, 256) -> RandAugment -> MultiScaleCrop +
# Resize(224) -> Normalize + background mix, 32 clips per batch (RawFrameClipLoader), frames in a temporary directory (page cache)
import shutil
import tempfile

from bdvcil_amd.decode import RawFrameClipLoader

root = tempfile.mkdtemp(prefix='bdv_frames_')
try:
    infos = []
    for v in range(32):
        d = os.path.join(root, f'v_{v}')
        os.makedirs(d)
        for i in range(1, 41):
            with open(os.path.join(d, f'img_{i:05}.jpg'), 'wb') as f:
                f.write(streams[(v * 7 + i) % 256])
        infos.append({'frame_dir': d, 'total_frames': 40, 'label': v % 10})
    bgs = []
    for k in range(8):
        p = os.path.join(root, f'bg_{k}.jpg')
        with open(p, 'wb') as f:
            f.write(streams[k])
        bgs.append(p)
    for threads in (4, 8, 16):
        loader = RawFrameClipLoader(dev, bg_files=bgs, threads=threads)
        for phase in ('train', 'val', 'test'):
            loader(infos, phase)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                b = loader(infos, phase)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f'RawFrameClipLoader {phase:5s} pipeline, {threads:2d} host threads: {dt * 1e3:6.1f} ms per batch of 32 clips = {32 / dt:5.0f} clips/s '
                  f'-> imgs {tuple(b["imgs"].shape)}')
finally:
    shutil.rmtree(root, ignore_errors=True)
