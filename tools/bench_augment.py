"""RandAugment kernels: time per operation slot for a batch of 32 clips x 8 frames x 256 x 340 (the frame size after
Resize(-1, 256) on UCF101), one operation for all clips, and for a drawn mix (prob 0.75, n = 2, m = 10)."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdvcil_amd import augment as A  # noqa: E402
from bdvcil_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device('cuda:0')
    B, T, H, W = 32, 8, 256, 340
    x = torch.randint(40, 220, (B, T, H, W, 3), dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    nbytes = 2 * x.numel()
    print(f'batch {B}x{T}x{H}x{W}x3 uint8 = {x.numel() / 1e6:.1f} MB; read + write = {nbytes / 1e6:.1f} MB per slot')
    for name, lo, hi in A.augment_list():
        val = (10.0 / 30) * float(hi - lo) + lo
        ri, rd = A.op_row(name, val, False, (170.0, 128.0), H, W)
        oi = torch.tensor([ri] * B, dtype=torch.int32, device=dev)
        od = torch.tensor([rd] * B, dtype=torch.float64, device=dev)
        ms = timeit(lambda: K.randaug_apply(x, oi, od, out=out))
        print(f'{name:13s} {ms:7.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s')
    aug = A.RandAugment(2, 10, 0.75)
    random.seed(0)
    np.random.seed(0)
    draws = [aug.draw(H, W) for _ in range(B)]
    rows = [(oi.to(dev), od.to(dev)) for oi, od in aug.rows(draws, H, W)]
    tmp = torch.empty_like(x)

    def both():
        K.randaug_apply(x, rows[0][0], rows[0][1], out=tmp)
        K.randaug_apply(tmp, rows[1][0], rows[1][1], out=out)
    ms = timeit(both)
    print(f'drawn mix, 2 slots: {ms:7.3f} ms per batch = {B / ms * 1e3:9.0f} clips/s  ({2 * nbytes / ms / 1e6:.1f} GB/s)')
    t0 = torch.cuda.Event(enable_timing=True)
    import time
    t = time.perf_counter()
    for _ in range(20):
        aug.rows([aug.draw(H, W) for _ in range(B)], H, W)
    print(f'host side (draws + tables) per batch: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms')


if __name__ == '__main__':
    main()
