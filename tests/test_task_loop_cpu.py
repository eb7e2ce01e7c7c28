"""Host-side bookkeeping of the CIL task loop (bdvcil_amd/task_loop.py): class bookkeeping, annotation / exemplar file
formats, epoch partitioning and the accuracy tables.  No GPU, no kernels."""
import json
import os

import pytest
import torch

from bdvcil_amd import task_loop as TL

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_task_splits_bookkeeping():
    # the UCF101 10-stage split shape of the configs: 51 + 10 x 5, original ids in arbitrary order
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(101, generator=g).tolist()
    splits = [perm[:51]] + [perm[51 + 5 * i:56 + 5 * i] for i in range(10)]
    ts = TL.TaskSplits(splits)
    assert ts.accumulate_task_size_list == [51 + 5 * i for i in range(11)]
    assert ts.num_classes(0) == 51 and ts.num_classes(10) == 101 and ts.num_classes(-1) == 101
    # incremental index = order of first appearance
    assert [ts.ori_idx_to_inc_idx[c] for c in perm] == list(range(101))
    assert ts.class_indices(3) == list(range(61, 66))


def _write_ann(path, rows):
    with open(path, 'w') as f:
        for r in rows:
            f.write('{} {} {}\n'.format(*r))


def test_annotation_files_round_trip(tmp_path):
    splits = TL.TaskSplits([[7, 2], [5], [0, 9]])
    train = [('a/v1', 30, 7), ('a/v2', 12, 2), ('b/v3', 44, 5), ('c/v4', 9, 0), ('c/v5', 10, 9), ('c/v6', 11, 3), ('a/v7', 5, 7)]
    val = [('a/w1', 3, 2), ('b/w2', 4, 5)]                 # no validation video of task 2 -> no file for it
    _write_ann(tmp_path / 'train.txt', train)
    _write_ann(tmp_path / 'val.txt', val)
    files = TL.CILWorkDir(tmp_path / 'work', splits)
    files.generate_annotation_file(tmp_path / 'train.txt', tmp_path / 'val.txt')
    names = [p.name for p in files.task_splits_ann_files['train']]
    assert names == ['train_task_0.txt', 'train_task_1.txt', 'train_task_2.txt']
    assert [p.name for p in files.task_splits_ann_files['val']] == ['val_task_0.txt', 'val_task_1.txt']
    # labels are rewritten to incremental ids, file order is preserved, class 3 (in no task) is dropped
    assert TL.read_ann_file(files.task_splits_ann_files['train'][0]) == [('a/v1', 30, 0), ('a/v2', 12, 1), ('a/v7', 5, 0)]
    assert TL.read_ann_file(files.task_splits_ann_files['train'][2]) == [('c/v4', 9, 3), ('c/v5', 10, 4)]
    with open(files.task_splits_ann_files['train'][1]) as f:
        assert f.read() == 'b/v3 44 2\n'
    other = TL.CILWorkDir(tmp_path / 'work', splits)
    other.collect_ann_files_from_work_dir(3)
    assert other.task_splits_ann_files['train'] == files.task_splits_ann_files['train']
    rec = TL.RawframeRecords(str(files.task_splits_ann_files['train'][0]), str(tmp_path))
    assert rec.video_infos[2] == dict(frame_dir=os.path.join(os.path.realpath(tmp_path), 'a/v7'), total_frames=5, label=0)


def test_exemplar_files(tmp_path):
    root = tmp_path / 'raw'
    root.mkdir()
    link = tmp_path / 'link'
    os.symlink(root, link)                                 # data_root may be a symlink: paths are compared after realpath
    splits = TL.TaskSplits([[0, 1], [2]])
    files = TL.CILWorkDir(tmp_path / 'work', splits)
    real = os.path.realpath(root)
    meta0 = {0: {'frame_dir': [f'{real}/k/x1', f'{real}/k/x2'], 'total_frames': torch.tensor([31, 7])},
             1: {'frame_dir': [f'{real}/m/y1'], 'total_frames': torch.tensor([5])}}
    meta1 = {2: {'frame_dir': [f'{real}/n/z1'], 'total_frames': torch.tensor([64])}}
    p0 = files.create_exemplar_ann_file(meta0, 0, str(link))
    files.create_exemplar_ann_file(meta1, 1, str(link))
    with open(p0) as f:
        assert f.read() == 'k/x1 31 0\nk/x2 7 0\nm/y1 5 1\n'
    tmp = files.combine_all_exemplar_ann_files(1)
    with open(tmp) as f:
        assert f.read() == 'k/x1 31 0\nk/x2 7 0\nm/y1 5 1\nn/z1 64 2'
    ex = TL.RawframeRecords(str(tmp), str(link))
    assert [v['label'] for v in ex.video_infos] == [0, 0, 1, 2] and ex.video_infos[3]['frame_dir'] == f'{real}/n/z1'
    assert files.ckpt_file(4).name == 'ckpt_task_4.pt' and files.class_mean_file(2).name == 'exemplar_class_mean_task_2.pt'
    with pytest.raises(ValueError):                        # an exemplar outside data_root cannot be written relative to it
        files.create_exemplar_ann_file({0: {'frame_dir': ['/elsewhere/v'], 'total_frames': torch.tensor([1])}}, 0, str(link))


def test_epoch_batches_match_torch_samplers():
    from torch.utils.data import DataLoader, DistributedSampler
    data = list(range(23))
    # one rank: DataLoader(shuffle=False/True, drop_last=False)
    assert TL.epoch_batches(23, 5, False) == [b.tolist() for b in DataLoader(data, batch_size=5, shuffle=False)]
    g1, g2 = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
    mine = TL.epoch_batches(23, 5, True, g1)
    assert sorted(sum(mine, [])) == data and mine == [torch.tensor(x).tolist() for x in mine]
    assert [len(b) for b in mine] == [5, 5, 5, 5, 3]
    assert TL.epoch_batches(23, 5, True, g2) == mine       # same generator state -> same permutation on every rank
    # several ranks: DistributedSampler's pad-and-stride partition of the same order
    for world in (2, 3, 4, 8, 32):
        covered = []
        for rank in range(world):
            ref = list(DistributedSampler(data, num_replicas=world, rank=rank, shuffle=False))
            got = sum(TL.epoch_batches(23, 4, False, None, rank, world), [])
            assert got == ref
            covered += got
        assert set(covered) == set(data)
    assert TL.epoch_batches(0, 4, True, torch.Generator().manual_seed(0)) == []


def test_accuracy_tables_match_reference_golden():
    with open(os.path.join(GOLDEN, 'table_golden.json')) as f:
        cases = json.load(f)
    assert len(cases) == 3
    for c in cases:
        meters = []
        for task in c['updates']:
            m = TL.AverageMeter()
            for val, n in task:
                m.update(val, n)
            meters.append(m)
        assert [m.avg for m in meters] == c['avg'] and [m.sum for m in meters] == c['sum']
        assert [m.count for m in meters] == c['count']
        assert TL.print_mean_accuracy(meters, c['classes']) == c['table']


def test_task_accuracies_slices():
    preds = torch.tensor([0, 1, 1, 0, 2, 2, 3, 0])
    labels = torch.tensor([[0], [1], [0], [0], [2], [3], [3], [3]])
    m = TL.task_accuracies(preds, labels, [4, 3])          # the 8th sample lies beyond the listed tasks, as in the reference
    assert m.values == [75.0, pytest.approx(200.0 / 3)] and m.sizes == [4, 3]
    assert m.avg == pytest.approx((75.0 * 4 + 200.0 / 3 * 3) / 7)


def test_attrdict_config():
    cfg = TL.AttrDict(dict(a=1, data=dict(features_extraction_epochs=1), optimizer=dict(type='SGD', paramwise_cfg=dict(k=5.0))))
    assert cfg.a == 1 and cfg.data.features_extraction_epochs == 1 and cfg.optimizer.paramwise_cfg.k == 5.0
    assert cfg.get('missing', 7) == 7 and 'a' in cfg and 'kd_modules_names' not in cfg
    with pytest.raises(AttributeError):
        cfg.nope
    import copy
    c2 = copy.deepcopy(cfg)
    c2.optimizer.type = 'X'
    assert cfg.optimizer.type == 'SGD'
