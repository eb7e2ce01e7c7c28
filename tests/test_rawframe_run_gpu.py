"""The whole chain on files: a two-task CIL run of ``CILTaskLoop`` whose ``clip_loader`` is ``RawFrameClipLoader`` -- JPEG frames on
disk -> host Huffman stage -> GPU decode -> Resize -> RandAugment -> MultiScaleCrop -> background mix -> TSM-R18 training, exemplar
selection from extracted features, CNN + NME testing through the TenCrop pipeline.  What a user of the reference does with
``cil_tools/train_cil.py`` on a rawframe dataset, at toy size.  Checks that the run completes, leaves the reference's files, and that
every number in them is finite; the stages themselves are tested against their oracles in test_jpeg_gpu.py / test_frames_gpu.py."""
import math
import re

import numpy as np
import pytest
import torch

from test_task_loop_gpu import _config

pytestmark = pytest.mark.gpu


def test_two_tasks_from_jpeg_files(tmp_path):
    import random
    from PIL import Image
    import bdvcil_amd.task_loop as TL
    from bdvcil_amd.decode import RawFrameClipLoader
    cfg = _config(tmp_path, task_splits=[[0, 1], [2, 3]], ending_task=1, num_epochs_per_task=2, videos_per_gpu=4, testing_videos_per_gpu=4)
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:60, 0:80]
    for name in ('train', 'val'):
        for rec in TL.read_ann_file(cfg[f'{name}_ann_file']):
            frame_dir, total, label = rec[0], int(rec[1]), int(rec[2])
            if label > 3:
                continue
            d = tmp_path / 'rawframes' / frame_dir
            d.mkdir(parents=True, exist_ok=True)
            base = np.stack([128 + 90 * np.sin((label + 1) * xx / 9.0), 128 + 90 * np.cos((label % 2 + 1) * yy / 7.0),
                             np.full(xx.shape, 60.0 * label)], -1)
            for i in range(1, total + 1):
                frame = np.clip(np.roll(base, 2 * i, axis=1) + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8)
                Image.fromarray(frame).save(str(d / f'img_{i:05}.jpg'), quality=80, subsampling=2)
    bgs = []
    for k in range(3):
        p = tmp_path / f'bg_{k}.jpg'
        Image.fromarray(rng.integers(0, 256, (90, 120, 3)).astype(np.uint8)).save(str(p), quality=85)
        bgs.append(str(p))
    torch.manual_seed(7); random.seed(7); np.random.seed(7)
    loader = RawFrameClipLoader('cuda', short_edge=128, input_size=112, bg_files=bgs, bg_resize=128, test_crop=('TenCrop', 128), threads=4)
    batch = loader([dict(frame_dir=str(tmp_path / 'rawframes' / 'class0' / 'train_v0_0'), total_frames=20, label=0)], 'train')
    assert tuple(batch['imgs'].shape) == (1, 8, 3, 112, 112) and 'randAug' in batch
    loop = TL.CILTaskLoop(cfg, loader, device='cuda', seed=0, log=lambda *a: None)
    history = loop.train()
    assert [h['task'] for h in history] == [0, 1]
    work = tmp_path / 'work'
    for t in range(2):
        assert (work / 'ckpt' / f'ckpt_task_{t}.pt').exists() and (work / 'exemplar' / f'exemplar_task_{t}.txt').exists()
        assert (work / 'ckpt' / f'exemplar_class_mean_task_{t}.pt').exists()
        means = torch.load(work / 'ckpt' / f'exemplar_class_mean_task_{t}.pt', weights_only=True)
        vals = list(means.values()) if isinstance(means, dict) else [means]
        assert vals and all(torch.isfinite(torch.as_tensor(v)).all() for v in vals)
        assert all(math.isfinite(v) and 0.0 <= v <= 100.0 for v in history[t]['cnn'].values)
    # the re-test of every task's checkpoint (cil_tools/test_cil.py in the reference) through the TenCrop pipeline, CNN + NME
    tables = TL.CILTaskLoop(_config(tmp_path, task_splits=[[0, 1], [2, 3]], ending_task=1, testing_videos_per_gpu=4), loader, device='cuda',
                            seed=0, log=lambda *a: None).cil_testing(test_nme=True)
    assert 'task 1' in tables['cnn'] and 'task 1' in tables['nme']
    for name in ('cnn_result.txt', 'nme_result.txt'):
        text = (work / name).read_text()
        nums = [float(v) for v in re.findall(r'\d+\.\d+', text)]                    # the accuracies of the table
        assert nums and all(math.isfinite(v) and 0.0 <= v <= 100.0 for v in nums), text
