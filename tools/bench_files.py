"""Training from JPEG files: the BASELINE config-2 step (TSM-R50, 32 clips of 8 x 224 x 224, CE, SGD) fed by RawFrameClipLoader -- files ->
host Huffman stage -> GPU decode -> Resize -> RandAugment -> MultiScaleCrop -> background mix -- instead of a resident synthetic batch.
One Python thread: the loader's host work for batch i + 1 runs while the GPU still executes step i (everything is enqueued
asynchronously; the loader's only wait is for the upload of two batches ago).  Prints clips/s with and without the loader.  Dev tool.
    python tools/bench_files.py [--threads 8] [--steps 20] [--arith bf16x3]"""
import argparse
import io
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument('--threads', type=int, default=8)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=4)
ap.add_argument('--arith', default='bf16x3')
args = ap.parse_args()

import bdvcil_amd as bd
from bdvcil_amd import kernels as K
from bench import model_cfg

K.set_conv_arith(args.arith)
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:240, 0:320]
root = tempfile.mkdtemp(prefix='bdv_frames_')
try:
    infos = []
    for v in range(64):
        d = os.path.join(root, f'v_{v}')
        os.makedirs(d)
        base = np.stack([128 + 100 * np.sin(xx / (7.0 + v % 5) + yy / 13.0), 128 + 90 * np.cos(xx / 5.0 + v), 128 + 80 * np.sin(yy / 3.0 + xx / 11.0)], -1)
        for i in range(1, 33):
            a = np.clip(np.roll(base, 3 * i, axis=1) + rng.normal(0, 10, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(a).save(os.path.join(d, f'img_{i:05}.jpg'), quality=85, subsampling=2)
        infos.append({'frame_dir': d, 'total_frames': 32, 'label': v % 101})
    bgs = []
    for k in range(8):
        p = os.path.join(root, f'bg_{k}.jpg')
        Image.fromarray(rng.integers(0, 256, (256, 340, 3)).astype(np.uint8)).save(p, quality=85)
        bgs.append(p)
    torch.manual_seed(0)
    model = bd.build_model(model_cfg(50, 101, 'SimpleLinear', 'CrossEntropyLoss', 0.5)).to(dev)
    model.train()
    opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0),
                                         lr=0.01, momentum=0.9, weight_decay=1e-4))
    engine = bd.TrainEngine(model, opt, grad_clip=None)
    loader = bd.RawFrameClipLoader(dev, bg_files=bgs, threads=args.threads)

    def batch_of(i):
        return loader([infos[(32 * i + k) % 64] for k in range(32)], 'train')

    def run(with_loader):
        fixed = batch_of(0)
        for i in range(args.warmup):
            engine.step(batch_of(i) if with_loader else fixed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = engine.step(batch_of(i) if with_loader else fixed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 32 * args.steps / dt, 1e3 * dt / args.steps, float(out['loss_cls'].detach())
    def run_prefetch():
        pre = bd.PrefetchLoader(loader, depth=2)
        lists = [[infos[(32 * i + k) % 64] for k in range(32)] for i in range(args.warmup + args.steps)]
        t0, n = None, 0
        for i, batch in enumerate(pre.iterate(lists, 'train')):
            if i == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = engine.step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 32 * args.steps / dt, 1e3 * dt / args.steps, float(out['loss_cls'].detach())
    a = run(False)
    b = run(True)
    c = run_prefetch()
    print(f'{args.arith}: resident batch {a[0]:.1f} clips/s ({a[1]:.2f} ms/step); from JPEG files through RawFrameClipLoader ({args.threads} host threads) '
          f'{b[0]:.1f} clips/s ({b[1]:.2f} ms/step) = {100 * b[0] / a[0]:.1f} % ; with PrefetchLoader (one batch ahead, own thread and stream) {c[0]:.1f} clips/s ({c[1]:.2f} ms/step) = {100 * c[0] / a[0]:.1f} % ; final loss {c[2]:.4f}')
finally:
    shutil.rmtree(root, ignore_errors=True)
