"""The class-incremental task loop without Lightning / mmcv.

Mirrors ``CILDataModule`` + ``CILTrainer`` of the reference (libs/cil/cil.py:28-375, :622-1140) for the parts a run on
the HIP path needs: task bookkeeping, the annotation / exemplar / checkpoint files on disk (same names, same text and
``torch.save`` formats, so a reference work_dir can be resumed here and vice versa), per-task optimizer rebuild, the
train / class-balanced-finetune phases, herding exemplar construction, exemplar class means, CNN + NME testing and
the model hand-over between tasks.  What it does not contain: decoding.  Frames come from a ``clip_loader`` callable
(the JPEG/RandAugment data path is SURVEY section 8(f) rank 3); ``SyntheticClipLoader`` below is a deterministic
stand-in with the same batch layout for tests and dry runs.

All tensor arithmetic goes through the HIP-backed modules of this package (``cil_step``, ``representation``,
``optim``); this file is bookkeeping.
"""
from __future__ import annotations

import copy
import os
import os.path as osp
import pathlib
import zlib
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist

from .cil_step import base_training_step, icarl_training_step, icarl_video_mix_training_step
from .ddp import GradAllReducer, broadcast_parameters
from .hooks import OutputHook
from .optim import build_lr_scheduler, build_optimizer
from .registry import build_model
from .representation import Herding, ReprPredictor, class_means_from_repr, nme_classify


class AttrDict(dict):
    """Attribute access over a (nested) config dict, the slice of ``mmcv.Config`` the loop touches."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                self[k] = AttrDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


class AverageMeter:
    """Per-task accuracies of one evaluation round with their sample counts (the record ``libs/utils.py:8-26`` keeps):
    ``values`` in the order they were added, ``avg`` their sample-weighted mean."""

    def __init__(self):
        self.values: List[float] = []
        self.sizes: List[int] = []

    def reset(self):
        del self.values[:], self.sizes[:]

    def update(self, val, n=1):
        self.values.append(val)
        self.sizes.append(n)

    @property
    def count(self):
        return sum(self.sizes)

    @property
    def sum(self):
        total = 0
        for v, n in zip(self.values, self.sizes):       # same accumulation order as a running sum
            total += v * n
        return total

    @property
    def avg(self):
        return self.sum / self.count if self.sizes else 0


def print_mean_accuracy(accuracies: List[AverageMeter], num_classes_per_task, floatfmt='.2f') -> str:
    """The accuracy table written to ``cnn_result.txt`` / ``nme_result.txt``, in the layout of libs/utils.py:29-48: one
    column per task's class range, one row per evaluated task (blank where a task did not exist yet), the weighted
    average last, and a final row with the mean of the averages."""
    from tabulate import tabulate
    num_tasks = len(num_classes_per_task)
    if len(accuracies) != num_tasks:
        raise AssertionError(f'{len(accuracies)} accuracy rows for {num_tasks} tasks')
    bounds = np.concatenate([[0], np.cumsum(num_classes_per_task)])
    headers = ['range'] + [f'{int(lo)}-{int(hi) - 1}' for lo, hi in zip(bounds[:-1], bounds[1:])] + ['Avg']
    rows = []
    for t, meter in enumerate(accuracies):
        cells = list(meter.values) + [None] * (num_tasks - 1 - t)
        rows.append([f'task {t}'] + cells + [meter.avg])
    rows.append(['avg_acc'] + [None] * num_tasks + [np.mean([m.avg for m in accuracies])])
    return tabulate(rows, headers=headers, floatfmt=[floatfmt] * 8, missingval='')


# ---- task splits and the files on disk --------------------------------------------------------------------------------

class TaskSplits:
    """Class bookkeeping of ``CILDataModule.__init__`` (libs/cil/cil.py:36-50)."""

    def __init__(self, task_splits: Sequence[Sequence[int]]):
        self.task_splits = [list(t) for t in task_splits]
        self.accumulate_task_size_list, acc = [], 0
        for t in self.task_splits:
            acc += len(t)
            self.accumulate_task_size_list.append(acc)
        self.ori_idx_to_inc_idx: Dict[int, int] = {}
        for t in self.task_splits:
            for i in t:
                if i not in self.ori_idx_to_inc_idx:
                    self.ori_idx_to_inc_idx[i] = len(self.ori_idx_to_inc_idx)

    def num_classes(self, task_idx: int) -> int:
        return self.accumulate_task_size_list[task_idx]        # task_idx = -1 -> total, as in the reference

    def class_indices(self, task_idx: int) -> List[int]:
        return [self.ori_idx_to_inc_idx[i] for i in self.task_splits[task_idx]]


def read_ann_file(path) -> List[Tuple[str, int, int]]:
    """``<frame_dir> <total_frames> <label>`` per line (RawframeDataset annotation format)."""
    out = []
    with open(path, 'r') as f:
        for line in f:
            if not line.strip():
                continue
            video_path, total_frames, label = line.strip().split()
            out.append((video_path, int(total_frames), int(label)))
    return out


class RawframeRecords:
    """The slice of mmaction's ``RawframeDataset`` the loop touches: ``video_infos`` (dicts with the absolute
    ``frame_dir``, ``total_frames``, ``label``), ``len()`` and ``bg_files`` for the background-mix variant."""

    def __init__(self, ann_file: Optional[str], data_prefix: str, test_mode: bool = False, phase: str = 'train'):
        self.ann_file = ann_file
        self.data_prefix = osp.realpath(data_prefix) if data_prefix is not None else None
        self.test_mode = test_mode
        self.phase = phase
        self.video_infos: List[dict] = []
        self.bg_files: List[str] = []
        if ann_file:
            for rel, total, label in read_ann_file(ann_file):
                frame_dir = osp.join(self.data_prefix, rel) if self.data_prefix is not None else rel
                self.video_infos.append(dict(frame_dir=frame_dir, total_frames=total, label=label))

    def __len__(self):
        return len(self.video_infos)

    def extend(self, others: Union['RawframeRecords', Iterable['RawframeRecords']]):
        """``CILDataModule.merge_dataset`` (libs/cil/cil.py:377-407): video_infos are appended in order."""
        for o in ([others] if isinstance(others, RawframeRecords) else others):
            self.video_infos.extend(o.video_infos)
        return self


class CILWorkDir:
    """File layout of a CIL run (libs/cil/cil.py:52-54, :84-125, :343-361, :305-316, :649-650, :616, :1061):

    ``task_splits/<template>.format(train|val, i)``   per-task annotation files with incremental labels
    ``exemplar/exemplar_task_<i>.txt``                ``<relative frame_dir> <total_frames> <class>`` per exemplar
    ``exemplar/tmp_exemplars.txt``                    concatenation of tasks 0..i
    ``ckpt/ckpt_task_<i>.pt``                         plain ``state_dict``
    ``ckpt/exemplar_class_mean_task_<i>.pt``          ``{'class_means': (K, D)}``
    """

    def __init__(self, work_dir, splits: TaskSplits, cil_ann_file_template: str = '{}_task_{}.txt'):
        self.work_dir = pathlib.Path(work_dir)
        self.splits = splits
        self.template = cil_ann_file_template
        self.work_dir.mkdir(exist_ok=True, parents=True)
        self.exemplar_dir = self.work_dir / 'exemplar'
        self.exemplar_dir.mkdir(exist_ok=True, parents=True)
        self.ckpt_dir = self.work_dir / 'ckpt'
        self.ckpt_dir.mkdir(exist_ok=True, parents=True)
        self.task_splits_ann_files = {'train': [], 'val': []}

    def generate_annotation_file(self, train_ann_file, val_ann_file) -> None:
        destination = self.work_dir / 'task_splits'
        destination.mkdir(exist_ok=True, parents=True)
        for train_val, file_path in zip(['train', 'val'], [train_ann_file, val_ann_file]):
            annotation_ = {}
            with open(file_path, 'r') as f:
                for line in f.readlines():
                    video_path, total_frames, label = line.strip().split()
                    annotation_[video_path] = total_frames, int(label)         # later duplicates win, as in the reference
            for task_i, class_indices in enumerate(self.splits.task_splits):
                wanted = set(class_indices)
                rows = ['{} {} {}\n'.format(v, tf, self.splits.ori_idx_to_inc_idx[lab])
                        for v, (tf, lab) in annotation_.items() if lab in wanted]
                if rows:
                    path = destination / self.template.format(train_val, task_i)
                    with open(path, 'w') as f:
                        f.writelines(rows)
                    self.task_splits_ann_files[train_val].append(path)

    def collect_ann_files_from_work_dir(self, num_tasks: int):
        d = self.work_dir / 'task_splits'
        for task_i in range(num_tasks):
            self.task_splits_ann_files['train'].append(d / self.template.format('train', task_i))
            self.task_splits_ann_files['val'].append(d / self.template.format('val', task_i))

    def exemplar_ann_file(self, task_idx: int) -> pathlib.Path:
        return self.exemplar_dir / 'exemplar_task_{}.txt'.format(task_idx)

    def create_exemplar_ann_file(self, exemplar_meta: dict, task_idx: int, data_root: str) -> str:
        root_dir = pathlib.Path(osp.realpath(data_root))
        ann_file = self.exemplar_ann_file(task_idx)
        with open(ann_file, 'w') as f:
            for class_idx, meta in exemplar_meta.items():
                for frame_dir, total_frames in zip(meta['frame_dir'], meta['total_frames']):
                    diff = pathlib.Path(frame_dir).relative_to(root_dir.absolute())
                    # the reference formats a 0-d tensor here; ``int`` prints the same digits
                    f.write('{} {} {}\n'.format(str(diff), int(total_frames), class_idx))
        return str(ann_file)

    def combine_all_exemplar_ann_files(self, task_idx: int) -> pathlib.Path:
        raw = []
        for i in range(task_idx + 1):
            with open(self.exemplar_ann_file(i), 'r') as f:
                raw.append(f.read().strip())
        tmp = self.exemplar_dir / 'tmp_exemplars.txt'
        with open(tmp, 'w') as f:
            f.write('\n'.join(raw))
        return tmp

    def ckpt_file(self, task_idx: int) -> pathlib.Path:
        return self.ckpt_dir / 'ckpt_task_{}.pt'.format(task_idx)

    def class_mean_file(self, task_idx: int) -> pathlib.Path:
        return self.ckpt_dir / 'exemplar_class_mean_task_{}.pt'.format(task_idx)


# ---- batches ----------------------------------------------------------------------------------------------------------

def epoch_batches(n: int, batch_size: int, shuffle: bool, generator: Optional[torch.Generator] = None, rank: int = 0,
                  world: int = 1) -> List[List[int]]:
    """Index batches of one epoch.  One rank: ``DataLoader(shuffle=..., drop_last=False)``.  Several ranks: the
    ``DistributedSampler`` partition Lightning installs under ``ddp_spawn`` (pad by wrapping to a multiple of the
    world size, then rank r takes positions r, r+world, ...)."""
    order = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
    if world > 1 and n > 0:
        total = -(-n // world) * world
        while len(order) < total:
            order += order[:total - len(order)]
        order = order[rank:total:world]
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]


class SyntheticClipLoader:
    """Deterministic stand-in for the decode pipeline with the reference's collated batch layout.

    Every video is a fixed uint8 clip derived from a CRC of its ``frame_dir`` plus a class-dependent low-frequency
    pattern (so a few epochs separate the classes); training batches go through ``TrainClipFrontEnd`` (RandAugment with
    probability ``randAug_prob``, background mix for the samples it skipped, comix_loader.py:105-116), all other phases
    through normalize only.  Keys: ``imgs`` (B, T, 3, H, W), ``label`` (B, 1) and the meta data
    ``frame_dir`` / ``total_frames`` / ``clip_len`` / ``num_clips`` / ``frame_inds`` that ``predict_step`` passes through."""

    def __init__(self, device, num_segments: int = 8, size: int = 64, randAug_prob: float = 0.75, alpha: float = 0.5, seed: int = 0):
        from .augment import RandAugment
        from .frontend import BackgroundMixFrontEnd, TrainClipFrontEnd
        self.device = torch.device(device)
        self.T, self.size, self.seed = num_segments, size, seed
        self.front = BackgroundMixFrontEnd(alpha=alpha)
        # the train pipeline of the configs: RandAugment(n=2, m=10, prob), background mix when it did not fire
        self.train_front = TrainClipFrontEnd(RandAugment(2, 10, randAug_prob), alpha=alpha, with_randAug=True)
        self._draws = 0

    def _frames(self, info: dict) -> np.ndarray:
        key = zlib.crc32(osp.basename(info['frame_dir']).encode()) ^ self.seed
        rng = np.random.default_rng(key)
        S, lab = self.size, info['label']
        yy, xx = np.mgrid[0:S, 0:S].astype(np.float32) / S
        base = np.stack([np.sin(2 * np.pi * ((lab % 5 + 1) * xx + 0.13 * lab)),
                         np.cos(2 * np.pi * ((lab // 5 + 1) * yy + 0.29 * lab)),
                         np.sin(2 * np.pi * ((lab % 3 + 1) * (xx + yy)))], axis=-1)          # (S, S, 3) in [-1, 1]
        frames = 128 + 70 * base[None] + 25 * rng.standard_normal((self.T, S, S, 3)).astype(np.float32)
        return np.clip(frames, 0, 255).astype(np.uint8)

    def __call__(self, video_infos: List[dict], phase: str) -> Dict:
        B = len(video_infos)
        frames = torch.from_numpy(np.stack([self._frames(v) for v in video_infos])).to(self.device)       # (B,T,S,S,3)
        if phase == 'train':
            g = torch.Generator().manual_seed(self.seed * 7919 + self._draws)
            self._draws += 1
            bg = torch.randint(0, 256, (B, self.size, self.size, 3), generator=g, dtype=torch.uint8).to(self.device)
            imgs, rand_flags, _ = self.train_front(frames, bg, as_nchw=True)
        else:
            imgs, rand_flags = self.front.as_nchw(frames), None
        tf = torch.tensor([v['total_frames'] for v in video_infos], dtype=torch.int64)
        dev = self.device                       # Lightning moves every tensor of the batch to the device
        extra = {} if rand_flags is None else {'randAug': rand_flags}       # Collect(keys=['imgs', 'label', 'randAug'])
        return {
            **extra,
            'imgs': imgs,
            'label': torch.tensor([[v['label']] for v in video_infos], dtype=torch.int64, device=self.device),
            'frame_dir': [v['frame_dir'] for v in video_infos],
            'total_frames': tf.to(dev),
            'clip_len': torch.ones(B, dtype=torch.int64, device=dev),
            'num_clips': torch.full((B,), self.T, dtype=torch.int64, device=dev),
            'frame_inds': torch.stack([torch.linspace(1, max(int(t), 1), self.T).long() for t in tf]).to(dev),
        }


# ---- the loop ---------------------------------------------------------------------------------------------------------

def task_accuracies(preds: torch.Tensor, labels: torch.Tensor, sizes: Sequence[int]) -> AverageMeter:
    """Per-task micro accuracy in percent over consecutive, un-shuffled slices (libs/cil/cil.py:936-941; torchmetrics
    multiclass ``Accuracy`` with its default micro average = fraction of correct samples)."""
    meter, start = AverageMeter(), 0
    preds, labels = preds.reshape(-1).cpu(), labels.reshape(-1).cpu()
    for n in sizes:
        acc = (preds[start:start + n] == labels[start:start + n]).float().mean().item() if n else float('nan')
        meter.update(acc * 100, n)
        start += n
    return meter


class CILTaskLoop:
    """``CILTrainer`` (libs/cil/cil.py:622-1140) on the HIP path.

    ``config`` carries the reference's top-level config keys (work_dir, task_splits, starting_task, ending_task,
    methods, num_epochs_per_task, videos_per_gpu, testing_videos_per_gpu, accumulate_grad_batches, optimizer,
    lr_scheduler, cbf_*, use_cbf, budget_size, storing_methods, budget_type, save_best, kd_modules_names,
    kd_weight_by_module, adaptive_scale_factors, repr_hook, data_root, train_ann_file, val_ann_file,
    cil_ann_file_template, model).  ``clip_loader(video_infos, phase) -> batch_data`` supplies the frames.
    """

    def __init__(self, config, clip_loader: Callable[[List[dict], str], Dict], device='cuda', seed: int = 0, log: Callable = print):
        self.config = config = config if isinstance(config, AttrDict) else AttrDict(config)
        self.device = torch.device(device)
        self.clip_loader = clip_loader
        self.log = log
        self.splits = TaskSplits(config.task_splits)
        self.files = CILWorkDir(config.work_dir, self.splits, config.get('cil_ann_file_template', '{}_task_{}.txt'))
        self.starting_task = config.get('starting_task', 0)
        self._current_task = self.starting_task
        self.ending_task = config.get('ending_task', len(self.splits.task_splits) - 1)
        self.num_tasks = min(len(self.splits.task_splits), self.ending_task + 1)
        self.method = config.get('methods', 'base')
        if self.method not in ('base', 'icarl', 'icarl_video_mix'):
            raise ValueError(self.method)
        self.use_kd = 'kd_modules_names' in config and self.method == 'base'
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._shuffle_gen = torch.Generator().manual_seed(seed)        # same permutation on every rank
        self.training_phase = None
        self.optimizer_mode = 'default'
        self.current_best = 0 if config.get('save_best', False) else None
        self.history: List[Dict] = []

        # models (libs/cil/cil.py:429-452): current + frozen previous copy, hooks for KD / representation
        self.current_model = build_model(config.model).to(self.device)
        self.prev_model = build_model(config.model).to(self.device)
        for p in self.prev_model.parameters():
            p.requires_grad_(False)
        self.prev_model.eval()
        if self.method in ('icarl', 'icarl_video_mix'):
            # libs/cil/icarl.py:34,:39: the iCaRL step trains on raw scores and feeds the teacher's scores to a softmax of
            # its own, whatever the config's test_cfg says (every shipped iCaRL config says 'prob')
            self.current_model.test_cfg['average_clips'] = 'score'
            self.prev_model.test_cfg['average_clips'] = 'score'
        broadcast_parameters(self.current_model)
        self.repr_module_name = config.get('repr_hook', 'cls_head.avg_pool')
        self.current_hooks = self.prev_hooks = None
        if self.use_kd:
            self.current_hooks = OutputHook(self.current_model, list(config.kd_modules_names), as_tensor=True)
            self.prev_hooks = OutputHook(self.prev_model, list(config.kd_modules_names), as_tensor=True)

        # data bookkeeping
        self.train_dataset: Optional[RawframeRecords] = None
        self.val_datasets: List[RawframeRecords] = []
        self.test_datasets: List[RawframeRecords] = []
        self.exemplar_datasets: List[RawframeRecords] = []
        if self.rank == 0:
            self.files.generate_annotation_file(config.train_ann_file, config.val_ann_file)
        self._barrier()
        if self.rank != 0:
            self.files.collect_ann_files_from_work_dir(len(self.splits.task_splits))
        if self.starting_task == 0:
            self.reload_train_dataset(use_internal_exemplar=False)
        else:
            self._resume()
        for i in range(self.num_tasks):
            self.val_datasets.append(RawframeRecords(str(self.files.task_splits_ann_files['val'][i]), config.data_root,
                                                     test_mode=True, phase='val'))

    # -- small helpers ---------------------------------------------------------------------------------------------------
    def _barrier(self):
        if self.world > 1:
            dist.barrier()

    @property
    def current_task(self) -> int:
        return self._current_task

    def num_classes(self, task_idx: int) -> int:
        return self.splits.num_classes(task_idx)

    @property
    def exemplar_size(self) -> int:
        return sum(len(e) for e in self.exemplar_datasets)

    def reload_train_dataset(self, use_internal_exemplar: bool = True):
        self.train_dataset = RawframeRecords(str(self.files.task_splits_ann_files['train'][self._current_task]),
                                             self.config.data_root, phase='train')
        if use_internal_exemplar:
            self.train_dataset.extend(self.exemplar_datasets)

    def build_exemplar_from_current_task(self, exemplar_meta: dict):
        if self.rank == 0:
            self.files.create_exemplar_ann_file(exemplar_meta, self._current_task, self.config.data_root)
        self._barrier()
        self.exemplar_datasets.append(RawframeRecords(str(self.files.exemplar_ann_file(self._current_task)),
                                                      self.config.data_root, phase='train'))

    def _load_state(self, model, path):
        """Every rank reads a checkpoint rank 0 wrote: wait for the writer first (a collective on the GPU stream does not
        order host file I/O)."""
        self._barrier()
        model.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def _save_state(self, obj, path):
        """Rank-0 write that other ranks can never see half-done: temporary file, then an atomic rename."""
        if self.rank == 0:
            tmp = f'{path}.tmp{os.getpid()}'
            torch.save(obj, tmp)
            os.replace(tmp, path)

    def _resume(self):
        """libs/cil/cil.py:659-696: rebuild exemplars (from disk, or by re-extracting them), roll back one task to
        load its weights into both models, then grow both classifiers to the starting task."""
        for task_idx in range(self._current_task):
            if not self.files.exemplar_ann_file(task_idx).exists():
                break
            self.exemplar_datasets.append(RawframeRecords(str(self.files.exemplar_ann_file(task_idx)), self.config.data_root))
        if len(self.exemplar_datasets) < self.starting_task:
            # the reference needs the weights of that task to extract features; it uses whatever current_model holds
            for i in range(len(self.exemplar_datasets), self.starting_task):
                self._current_task = i
                self.current_model.update_fc(self.num_classes(i))
                self._load_state(self.current_model, self.files.ckpt_file(i))
                self.build_exemplar_from_current_task(self._construct_exemplar())
            self._current_task = self.starting_task
        self._current_task -= 1
        self.current_model.update_fc(self.num_classes(self._current_task))
        self._load_state(self.current_model, self.files.ckpt_file(self._current_task))
        self.prev_model.update_fc(self.num_classes(self._current_task))
        self.prev_model.load_state_dict(self.current_model.state_dict())
        self.prev_model.eval()
        self._current_task += 1
        self.current_model.update_fc(self.num_classes(self._current_task))
        self.prev_model.update_fc(self.num_classes(self._current_task))
        self._freeze_prev()
        self.reload_train_dataset(use_internal_exemplar=True)

    def _freeze_prev(self):
        for p in self.prev_model.parameters():
            p.requires_grad_(False)
        self.prev_model.eval()

    # -- one fit ---------------------------------------------------------------------------------------------------------
    def _training_step(self, batch_data: Dict) -> Dict:
        t = self._current_task
        if self.method == 'icarl_video_mix':
            loss = icarl_video_mix_training_step(self.current_model, batch_data, self.num_classes(t), self.config.video_mix_prob,
                                                 self.config.video_mix_alpha, current_task=t, prev_model=self.prev_model,
                                                 previous_task_num_classes=self.num_classes(t - 1))
            return {'loss': loss, 'loss_cls': loss}
        if self.method == 'icarl':
            loss = icarl_training_step(self.current_model, batch_data, self.num_classes(t), current_task=t,
                                       prev_model=self.prev_model, previous_task_num_classes=self.num_classes(t - 1))
            return {'loss': loss, 'loss_cls': loss}
        cfg = self.config
        return base_training_step(self.current_model, batch_data, current_task=t,
                                  prev_model=self.prev_model if self.use_kd else None,
                                  current_hooks=self.current_hooks, prev_hooks=self.prev_hooks,
                                  kd_modules_names=list(cfg.kd_modules_names) if self.use_kd else (),
                                  kd_weight_by_module=list(cfg.kd_weight_by_module) if self.use_kd else (),
                                  adaptive_scale_factors=list(cfg.adaptive_scale_factors) if self.use_kd else (),
                                  kd_exemplar_only=cfg.get('kd_exemplar_only', False),
                                  previous_task_num_classes=self.num_classes(t - 1))

    def fit(self, records: RawframeRecords, max_epochs: int, validate: bool = False) -> List[float]:
        """One ``pl.Trainer.fit`` (libs/cil/cil.py:745-760, :776-800): a fresh optimizer + scheduler from the config
        of the current ``optimizer_mode``, gradient clipping by global norm 1.0 after task 0, gradient accumulation
        with the loss divided by the accumulation count, scheduler stepped per epoch."""
        cfg = self.config
        opt_cfg, sch_cfg = ((cfg.optimizer, cfg.get('lr_scheduler')) if self.optimizer_mode == 'default'
                            else (cfg.cbf_optimizer, cfg.get('cbf_lr_scheduler')))
        self.current_model.train()
        optimizer = build_optimizer(self.current_model, dict(opt_cfg))
        scheduler = build_lr_scheduler(optimizer, sch_cfg) if sch_cfg else None
        reducer = None
        if self.world > 1:
            # DistributedDataParallel is constructed per fit and starts from rank 0's weights and buffers: this is what
            # makes the freshly initialised classifier rows of ``update_fc`` agree across ranks
            broadcast_parameters(self.current_model)
            reducer = GradAllReducer(self.current_model)
            optimizer.set_grad_scale(reducer.grad_scale)
        clip = None if self._current_task == 0 else 1.0
        accum = int(cfg.get('accumulate_grad_batches', 1))
        if accum > 1 and reducer is not None:
            raise NotImplementedError('accumulate_grad_batches > 1 with several ranks (the 8-GPU setting of the configs uses 1)')
        epoch_losses = []
        try:
            self._fit_epochs(records, max_epochs, validate, optimizer, scheduler, reducer, clip, accum, epoch_losses)
        finally:
            if reducer is not None:
                reducer.remove()        # also on an exception: stale post-accumulate hooks would fire collectives in the next fit
        if reducer is not None:
            # DDP keeps module buffers equal to rank 0's at every forward; do it once per fit here
            for b in self.current_model.buffers():
                dist.broadcast(b.data, 0)
        return epoch_losses

    def _fit_epochs(self, records, max_epochs, validate, optimizer, scheduler, reducer, clip, accum, epoch_losses):
        cfg = self.config
        for epoch in range(max_epochs):
            self.current_model.train()
            batches = epoch_batches(len(records), cfg.videos_per_gpu, True, self._shuffle_gen, self.rank, self.world)
            total, pending = 0.0, 0
            optimizer.zero_grad(set_to_none=True)
            for bi, idx in enumerate(batches):
                batch_data = self.clip_loader([records.video_infos[i] for i in idx], records.phase)
                losses = self._training_step(batch_data)
                (losses['loss'] / accum if accum > 1 else losses['loss']).backward()
                pending += 1
                total = total + losses['loss'].detach()
                if pending == accum or bi == len(batches) - 1:
                    if reducer is not None:
                        reducer.finish()
                    if clip:
                        optimizer.clip_grad_norm_(clip)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
                    pending = 0
            if scheduler is not None:
                scheduler.step()
            epoch_losses.append(float(total) / max(len(batches), 1))
            if validate:
                self._validation_epoch()

    def _validation_epoch(self):
        """``validation_step`` / ``validation_epoch_end`` (libs/cil/cil.py:580-618): CNN accuracy over the validation
        sets of tasks 0..current; the best weights go to ``ckpt_task_<i>.pt``."""
        records = self._merged(self.val_datasets[0:self._current_task + 1], 'val')
        pred = self.predict(records, self.config.get('testing_videos_per_gpu', 1), extract_repr=False)
        cls_score = torch.cat([p['cls_score'] for p in pred], dim=0)
        labels = torch.cat([p['label'] for p in pred], dim=0)
        acc = task_accuracies(torch.argmax(cls_score, dim=1), labels, [len(d) for d in self.val_datasets[:self._current_task + 1]])
        if self.current_best < acc.avg:
            self.log('Accuracy improve from {} to {}'.format(self.current_best, acc.avg))
            self.current_best = acc.avg
            self._save_state(self.current_model.state_dict(), self.files.ckpt_file(self._current_task))
        return acc

    def train_task(self) -> List[float]:
        self.training_phase = 'inc_step'
        validate = bool(self.config.get('save_best', False)) and self._current_task == 0
        if validate:
            self.current_best = 0
        return self.fit(self.train_dataset, self.config.num_epochs_per_task, validate)

    def build_cbf_dataset(self) -> RawframeRecords:
        """Class-balanced set = all exemplars so far (libs/cil/cil.py:160-187)."""
        return RawframeRecords(None, self.config.data_root, phase='train').extend(self.exemplar_datasets)

    def train_cbf(self) -> List[float]:
        self.training_phase = 'cbf_step'
        validate = bool(self.config.get('save_best', False))
        if validate:
            self.current_best = 0
        cbf = self.build_cbf_dataset()
        self.optimizer_mode = 'cbf'
        try:
            if self.config.get('cbf_train_backbone', False):
                return self.fit(cbf, self.config.cbf_num_epochs_per_task, validate)
            self.current_model.freeze_backbone()
            try:
                return self.fit(cbf, self.config.cbf_num_epochs_per_task, validate)
            finally:
                self.current_model.unfreeze_backbone()
        finally:
            self.optimizer_mode = 'default'

    # -- prediction ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _merged(datasets: Sequence[RawframeRecords], phase: str) -> RawframeRecords:
        out = RawframeRecords(None, None, test_mode=True, phase=phase)
        return out.extend(datasets)

    @torch.no_grad()
    def predict(self, records: RawframeRecords, batch_size: int, extract_repr: bool = True, extract_meta: bool = False) -> List[Dict]:
        """``CILTrainer.predict`` (libs/cil/cil.py:1091-1140) without the per-rank writer files: every rank runs the
        whole set (identical weights and buffers), so the collated result needs no gather."""
        was_training = self.current_model.training
        self.current_model.eval()
        predictor = ReprPredictor(self.current_model, self.repr_module_name, extract_repr=extract_repr, extract_meta=extract_meta)
        out = []
        try:
            for idx in epoch_batches(len(records), batch_size, False):
                out.append(predictor.predict_step(self.clip_loader([records.video_infos[i] for i in idx], records.phase)))
        finally:
            predictor.close()
            self.current_model.train(was_training)
        return out

    def _extract_features_for_constructing_exemplar(self) -> Dict:
        """libs/cil/cil.py:872-908."""
        records = RawframeRecords(str(self.files.task_splits_ann_files['train'][self._current_task]), self.config.data_root,
                                  test_mode=True, phase='features_extraction')
        pred_ = self.predict(records, self.config.videos_per_gpu, extract_repr=True, extract_meta=True)
        epochs = self.config.get('data', {}).get('features_extraction_epochs', 1)
        repr_ = torch.cat([b['mean_crops_repr_'] for b in pred_], dim=0)
        repr_ = repr_.reshape(-1, epochs, repr_.size(1))
        cls_score = torch.cat([b['cls_score'] for b in pred_], dim=0)
        cls_score = cls_score.reshape(-1, epochs, cls_score.size(1))
        return {
            'frame_dir': [fd for b in pred_ for fd in b['frame_dir']],
            'total_frames': torch.cat([b['total_frames'] for b in pred_], dim=0),
            'label': torch.cat([b['label'] for b in pred_], dim=0).squeeze(dim=1),
            'clip_len': torch.cat([b['clip_len'] for b in pred_], dim=0),
            'num_clips': torch.cat([b['num_clips'] for b in pred_], dim=0),
            'frame_inds': torch.cat([b['frame_inds'] for b in pred_], dim=0),
            'repr_': repr_,
            'cls_score': cls_score,
        }

    def _construct_exemplar(self) -> Dict:
        cfg = self.config
        manager = Herding(budget_size=cfg.budget_size, class_indices=self.splits.class_indices(self._current_task),
                          cosine_distance=True, storing_methods=cfg.get('storing_methods', 'videos'),
                          budget_type=cfg.get('budget_type', 'class'))
        return manager.construct_exemplar(self._extract_features_for_constructing_exemplar())

    def _get_exemplar_class_means(self, task_idx: int, override_class_mean_ckpt: bool = False) -> torch.Tensor:
        """libs/cil/cil.py:1056-1089."""
        path = self.files.class_mean_file(task_idx)
        if not override_class_mean_ckpt and path.exists():
            return torch.load(path, map_location=self.device, weights_only=True)['class_means']
        self.current_model.update_fc(self.num_classes(self._current_task))
        if self.rank == 0:
            self.files.combine_all_exemplar_ann_files(task_idx)
        self._barrier()
        records = RawframeRecords(str(self.files.exemplar_dir / 'tmp_exemplars.txt'), self.config.data_root, test_mode=True,
                                  phase='features_extraction')
        pred_ = self.predict(records, self.config.get('testing_videos_per_gpu', 1), extract_repr=True)
        repr_ = torch.cat([b['mean_crops_repr_'] for b in pred_], dim=0)
        label = torch.cat([b['label'] for b in pred_], dim=0).squeeze(dim=1)
        class_means = class_means_from_repr(repr_, label, self.num_classes(task_idx))
        self._save_state({'class_means': class_means}, path)
        self._barrier()
        return class_means

    def _testing(self, task_indices: Sequence[int], val_test: str = 'test', exemplar_class_means: Optional[torch.Tensor] = None):
        """libs/cil/cil.py:910-983.  Returns ``cnn_accuracies`` or ``(cnn_accuracies, nme_accuracies)``."""
        assert len(task_indices) == 2
        ds_list = self.val_datasets if val_test == 'val' or not self.test_datasets else self.test_datasets
        records = self._merged(ds_list[task_indices[0]:task_indices[1] + 1], val_test)
        pred_ = self.predict(records, self.config.get('testing_videos_per_gpu', 1), extract_repr=exemplar_class_means is not None)
        cls_score = torch.cat([b['cls_score'] for b in pred_], dim=0)
        labels = torch.cat([b['label'] for b in pred_], dim=0)
        sizes = [len(ds_list[i]) for i in range(self._current_task + 1)]
        cnn = task_accuracies(torch.argmax(cls_score, dim=1), labels, sizes)
        self.log('Task {} Accuracies (CNN): {}\nAvg Accuracy (CNN): {}'.format(self._current_task, cnn.values, cnn.avg))
        if exemplar_class_means is None:
            return cnn
        repr_ = torch.cat([b['repr_'] for b in pred_], dim=0)                    # (num_samples, num_crops, dim)
        _, preds_nme = nme_classify(repr_, exemplar_class_means)
        nme = task_accuracies(preds_nme, labels, sizes)
        self.log('Task {} Accuracies (NME): {}\nAvg Accuracy (NME): {}'.format(self._current_task, nme.values, nme.avg))
        return cnn, nme

    # -- the loop (libs/cil/cil.py:806-861) ------------------------------------------------------------------------------
    def train(self):
        cfg = self.config
        while self._current_task < self.num_tasks:
            t = self._current_task
            self.log('Task {}, current heads: {}\nTraining set size: {} (including {} samples from exemplar)'.format(
                t, self.num_classes(t), len(self.train_dataset), self.exemplar_size))
            record = {'task': t, 'train_loss': self.train_task()}
            save_best = bool(cfg.get('save_best', False))
            if save_best and t == 0:
                self._load_state(self.current_model, self.files.ckpt_file(t))
            self.build_exemplar_from_current_task(self._construct_exemplar())
            if t > 0 and cfg.get('use_cbf', False):
                record['cbf_loss'] = self.train_cbf()
            if save_best:
                self._load_state(self.current_model, self.files.ckpt_file(t))
            else:
                self._save_state(self.current_model.state_dict(), self.files.ckpt_file(t))
                self._barrier()
            class_means = self._get_exemplar_class_means(t, override_class_mean_ckpt=True)
            record['cnn'], record['nme'] = self._testing([0, t], val_test='val', exemplar_class_means=class_means)
            self.history.append(record)
            self._current_task += 1
            if self._current_task < self.num_tasks:
                self.prev_model.load_state_dict(self.current_model.state_dict())
                self.prev_model.eval()
                self.current_model.update_fc(self.num_classes(self._current_task))
                self.prev_model.update_fc(self.num_classes(self._current_task))
                self._freeze_prev()
                self.reload_train_dataset(use_internal_exemplar=True)
        return self.history

    def cil_testing(self, test_nme: bool = False):
        """libs/cil/cil.py:985-1030: re-test every task's checkpoint, write ``cnn_result.txt`` / ``nme_result.txt``."""
        tmp = self._current_task
        cnn_all, nme_all = [], []
        self.test_datasets = [RawframeRecords(str(self.files.task_splits_ann_files['val'][i]), self.config.data_root,
                                              test_mode=True, phase='test') for i in range(self.num_tasks)]
        for task_idx in range(self.num_tasks):
            self._current_task = task_idx
            self.current_model.update_fc(self.num_classes(task_idx))
            self._load_state(self.current_model, self.files.ckpt_file(task_idx))
            if test_nme:
                means = self._get_exemplar_class_means(task_idx, override_class_mean_ckpt=False)
                cnn, nme = self._testing([0, task_idx], exemplar_class_means=means)
                nme_all.append(nme)
            else:
                cnn = self._testing([0, task_idx])
            cnn_all.append(cnn)
        sizes = [len(c) for c in self.splits.task_splits[self.starting_task:self.ending_task + 1]]
        tables = {'cnn': print_mean_accuracy(cnn_all, sizes[:len(cnn_all)])}
        if self.rank == 0:
            with open(self.files.work_dir / 'cnn_result.txt', 'w') as f:
                f.write('CNN Accuracies' + tables['cnn'] + '\n')
        if test_nme:
            tables['nme'] = print_mean_accuracy(nme_all, sizes[:len(nme_all)])
            if self.rank == 0:
                with open(self.files.work_dir / 'nme_result.txt', 'w') as f:
                    f.write('NME Accuracies' + tables['nme'] + '\n')
        self._current_task = tmp
        return tables
