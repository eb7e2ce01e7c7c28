"""The CIL task loop end to end on the GPU (bdvcil_amd/task_loop.py): three tasks of a tiny
TSM-R18 on synthetic clips.  Checks the files a reference run leaves behind, the exemplar selection and class means
against the CPU oracle, resume (with and without the exemplar files) and the class-balanced fine-tuning phase."""
import copy
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TASKS = [[4, 1], [5, 0], [2, 3]]
N_TRAIN, N_VAL, BUDGET, SIZE = 6, 3, 2, 64


def _model_cfg(num_classes):
    return dict(type='CILRecognizer2D',
                backbone=dict(type='ResNetTSM', pretrained=None, depth=18, norm_eval=False, num_segments=8, shift_div=8),
                cls_head=dict(type='IncrementalTSMHead', num_classes=num_classes, in_channels=512,
                              inc_head_config=dict(type='LocalSimilarityClassifier', out_features=num_classes, nb_proxies=1),
                              num_segments=8, loss_cls=dict(type='LSCLoss'), spatial_type='avg',
                              consensus=dict(type='AvgConsensus', dim=1), dropout_ratio=0.5, init_std=0.001, is_shift=True),
                train_cfg=None, test_cfg=dict(average_clips='prob'))


def _config(tmp, **over):
    root = tmp / 'rawframes'
    root.mkdir(exist_ok=True)
    for name, n in (('train', N_TRAIN), ('val', N_VAL)):
        with open(tmp / f'{name}.txt', 'w') as f:
            for c in range(6):
                for k in range(n):
                    f.write(f'class{c}/{name}_v{c}_{k} {20 + 3 * k + c} {c}\n')
    opt = dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised', paramwise_cfg=dict(fc_lr_scale_factor=5.0),
               lr=0.01, momentum=0.9, weight_decay=0.0001)
    cfg = dict(work_dir=str(tmp / 'work'), task_splits=TASKS, methods='base', starting_task=0, ending_task=2,
               num_epochs_per_task=3, videos_per_gpu=6, testing_videos_per_gpu=4, accumulate_grad_batches=1,
               use_cbf=False, cbf_train_backbone=False, cbf_num_epochs_per_task=1, budget_size=BUDGET, storing_methods='videos',
               budget_type='class', save_best=False,
               kd_modules_names=['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool'],
               repr_hook='cls_head.avg_pool', kd_exemplar_only=False, kd_weight_by_module=[0.01] * 5,
               adaptive_scale_factors=[1.0, 1.4142, 1.7320],
               optimizer=opt, lr_scheduler=dict(type='MultiStepLR', params=dict(milestones=[2], gamma=0.1)),
               cbf_optimizer=dict(opt), cbf_lr_scheduler=dict(type='MultiStepLR', params=dict(milestones=[2], gamma=0.1)),
               data_root=str(root), train_ann_file=str(tmp / 'train.txt'), val_ann_file=str(tmp / 'val.txt'),
               cil_ann_file_template='{}_task_{}.txt', data=dict(features_extraction_epochs=1), model=_model_cfg(len(TASKS[0])))
    cfg.update(over)
    return cfg


def _loop(cfg, seed=0):
    import random
    import bdvcil_amd.task_loop as TL
    torch.manual_seed(1234)
    random.seed(1234)                  # RandAugment of the synthetic train loader draws from these two
    np.random.seed(1234)
    loader = TL.SyntheticClipLoader('cuda', num_segments=8, size=SIZE, seed=5)
    return TL.CILTaskLoop(cfg, loader, device='cuda', seed=seed, log=lambda *a: None)


@pytest.fixture(scope='module')
def finished_run(tmp_path_factory):
    tmp = tmp_path_factory.mktemp('cil_run')
    loop = _loop(_config(tmp))
    history = loop.train()
    return tmp, loop, history


def test_files_of_a_run(finished_run):
    import bdvcil_amd.task_loop as TL
    tmp, loop, history = finished_run
    work = tmp / 'work'
    assert [h['task'] for h in history] == [0, 1, 2] and loop.current_task == 3
    for t in range(3):
        K = 2 * (t + 1)
        assert TL.read_ann_file(work / 'task_splits' / f'train_task_{t}.txt')[0][2] in (2 * t, 2 * t + 1)
        ex = TL.read_ann_file(work / 'exemplar' / f'exemplar_task_{t}.txt')
        assert [r[2] for r in ex] == [2 * t] * BUDGET + [2 * t + 1] * BUDGET            # class order = task_splits order
        train_rows = {r[0]: r for r in TL.read_ann_file(work / 'task_splits' / f'train_task_{t}.txt')}
        assert all(train_rows[r[0]] == r for r in ex)                                     # relative path, frames, label survive
        sd = torch.load(work / 'ckpt' / f'ckpt_task_{t}.pt', weights_only=True)
        assert set(sd) == set(loop.current_model.state_dict())
        assert sd['cls_head.fc_cls.weights'].shape == (K, 512)
        cm = torch.load(work / 'ckpt' / f'exemplar_class_mean_task_{t}.pt', weights_only=True)
        assert set(cm) == {'class_means'} and cm['class_means'].shape == (K, 512)
        assert len(history[t]['cnn'].values) == t + 1 and history[t]['cnn'].sizes == [2 * N_VAL] * (t + 1)
        assert len(history[t]['train_loss']) == 3 and all(np.isfinite(history[t]['train_loss']))
    with open(work / 'exemplar' / 'tmp_exemplars.txt') as f:
        assert len(f.read().split('\n')) == 6 * BUDGET
    # the models after the last task: prev is a frozen copy of the model that finished task 1, grown to task 2's classes
    assert loop.current_model.cls_head.num_classes == 6
    assert not any(p.requires_grad for p in loop.prev_model.parameters()) and not loop.prev_model.training
    # task 0 is learnable from the class-dependent pattern: the loss goes down and the held-out clips are classified
    assert history[0]['train_loss'][-1] < history[0]['train_loss'][0]
    assert all(0.0 <= h[k].avg <= 100.0 for h in history for k in ('cnn', 'nme'))


def test_exemplars_and_class_means_match_oracle(finished_run):
    """Re-derive task 1's exemplar file and class-mean file from its checkpoint with the CPU oracle."""
    import bdvcil_amd.task_loop as TL
    from oracle import repr_oracle as RO
    tmp, loop, _ = finished_run
    work = tmp / 'work'
    probe = _loop(_config(tmp, work_dir=str(tmp / 'probe')))
    probe.files = loop.files
    probe._current_task = 1
    probe.current_model.update_fc(4)
    probe._load_state(probe.current_model, loop.files.ckpt_file(1))
    meta = probe._extract_features_for_constructing_exemplar()
    feats, labels = meta['repr_'].cpu(), meta['label'].cpu().numpy()
    assert tuple(feats.shape) == (2 * N_TRAIN, 1, 512) and sorted(set(labels.tolist())) == [2, 3]
    ex = TL.read_ann_file(work / 'exemplar' / 'exemplar_task_1.txt')
    root = os.path.realpath(tmp / 'rawframes')
    for ci, c in enumerate([2, 3]):
        rows = np.nonzero(labels == c)[0]
        _, idx, _ = RO.herding_select(feats[torch.from_numpy(rows), 0], BUDGET, True)
        want = [os.path.relpath(meta['frame_dir'][rows[i]], root) for i in idx]
        assert [r[0] for r in ex[ci * BUDGET:(ci + 1) * BUDGET]] == want
    # class means over the exemplars of tasks 0..1, features from the same checkpoint
    recs = TL.RawframeRecords(str(work / 'exemplar' / 'tmp_exemplars.txt'), str(tmp / 'rawframes'), phase='features_extraction')
    recs.video_infos = recs.video_infos[:4 * BUDGET]
    pred = probe.predict(recs, 4, extract_repr=True)
    r = torch.cat([p['mean_crops_repr_'] for p in pred]).cpu().numpy()
    lab = torch.cat([p['label'] for p in pred]).view(-1).cpu().numpy()
    want = np.stack([r[lab == k].astype(np.float64).mean(0) for k in range(4)])
    got = torch.load(work / 'ckpt' / 'exemplar_class_mean_task_1.pt', weights_only=True)['class_means'].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_resume_from_work_dir(finished_run, tmp_path):
    import bdvcil_amd.task_loop as TL
    tmp, loop, history = finished_run
    for name in ('train.txt', 'val.txt'):
        shutil.copy(tmp / name, tmp_path / name)
    shutil.copytree(tmp / 'work', tmp_path / 'work')
    os.remove(tmp_path / 'work' / 'exemplar' / 'exemplar_task_1.txt')          # must be re-derived from ckpt_task_1
    os.remove(tmp_path / 'work' / 'exemplar' / 'exemplar_task_2.txt')
    cfg = _config(tmp_path, starting_task=2)
    cfg['data_root'] = str(tmp / 'rawframes')                                   # same absolute frame_dirs -> same synthetic clips
    resumed = _loop(cfg)
    with open(tmp / 'work' / 'exemplar' / 'exemplar_task_1.txt') as a, open(tmp_path / 'work' / 'exemplar' / 'exemplar_task_1.txt') as b:
        assert a.read() == b.read()
    assert resumed.current_task == 2 and len(resumed.exemplar_datasets) == 2
    assert len(resumed.train_dataset) == 2 * N_TRAIN + 4 * BUDGET
    sd1 = torch.load(tmp / 'work' / 'ckpt' / 'ckpt_task_1.pt', weights_only=True)
    cur, prev = resumed.current_model.state_dict(), resumed.prev_model.state_dict()
    for k, v in sd1.items():
        if k.startswith('cls_head.fc_cls'):
            assert cur[k].shape[0] == 6 and torch.equal(cur[k][:4].cpu(), v.cpu()) and torch.equal(prev[k][:4].cpu(), v.cpu())
        else:
            assert torch.equal(cur[k].cpu(), v.cpu()) and torch.equal(prev[k].cpu(), v.cpu())
    hist = resumed.train()
    assert [h['task'] for h in hist] == [2] and len(hist[0]['cnn'].values) == 3
    assert (tmp_path / 'work' / 'exemplar' / 'exemplar_task_2.txt').exists()
    with pytest.raises(RuntimeError):                     # classifiers only grow (LSC.update_fc), as in the reference:
        resumed.cil_testing(test_nme=True)                # the per-task re-test starts from a task-0 sized model
    cfg0 = _config(tmp_path)
    cfg0['data_root'] = cfg['data_root']
    tables = _loop(cfg0).cil_testing(test_nme=True)
    with open(tmp_path / 'work' / 'cnn_result.txt') as f:
        assert f.read() == 'CNN Accuracies' + tables['cnn'] + '\n'
    assert 'task 2' in tables['nme'] and (tmp_path / 'work' / 'nme_result.txt').exists()
    with pytest.raises(FileNotFoundError):                                       # nothing to resume from
        _loop(_config(tmp_path, starting_task=1, work_dir=str(tmp_path / 'empty')))


def test_cbf_phase_and_icarl(tmp_path):
    (tmp_path / 'a').mkdir()
    cfg = _config(tmp_path / 'a', use_cbf=True, ending_task=1, num_epochs_per_task=1, accumulate_grad_batches=2)
    loop = _loop(cfg)
    seen = {}
    fit = loop.fit

    def spy(records, max_epochs, validate=False):
        before = copy.deepcopy(loop.current_model.backbone.state_dict())
        grads = [p.requires_grad for p in loop.current_model.backbone.parameters()]
        out = fit(records, max_epochs, validate)
        if loop.training_phase == 'cbf_step':
            seen['records'] = len(records)
            seen['frozen'] = not any(grads)
            after = loop.current_model.backbone.state_dict()
            seen['weights_kept'] = all(torch.equal(before[k], after[k]) for k in before if 'running' not in k and 'num_batches' not in k)
        return out
    loop.fit = spy
    hist = loop.train()
    assert seen == {'records': 4 * BUDGET, 'frozen': True, 'weights_kept': True}
    assert all(p.requires_grad for p in loop.current_model.backbone.parameters()) and loop.optimizer_mode == 'default'
    assert 'cbf_loss' in hist[1] and 'cbf_loss' not in hist[0]

    (tmp_path / 'b').mkdir()
    cfg = _config(tmp_path / 'b', methods='icarl', ending_task=1, num_epochs_per_task=1)
    cfg['model']['cls_head']['inc_head_config'] = dict(type='SimpleLinear', out_features=2)
    cfg['model']['cls_head']['loss_cls'] = dict(type='CrossEntropyLoss')
    # the shipped iCaRL configs say average_clips='prob' (kept here); the loop switches both models to 'score' (icarl.py:34,39)
    hist = _loop(cfg).train()
    assert len(hist) == 2 and all(np.isfinite(h['train_loss']).all() for h in hist)
