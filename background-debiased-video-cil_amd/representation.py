"""Inference + representation path of the CIL loop on the HIP kernels.

Mirrors, with the reference's names and result layouts:

* ``BaseCIL.predict_step`` with ``extract_repr`` / ``extract_meta``  (libs/cil/cil.py:558-578, ``_extract_repr`` :501-506)
* the NME cosine classifier of ``_testing``                            (libs/cil/cil.py:945-960)
* the class means of the exemplar representations                     (libs/cil/cil.py:1079-1083)
* ``Herding``                                                         (libs/cil/memory_selection.py:7-164)

Bookkeeping (splitting meta data by class, gathering the chosen rows) stays in Python exactly as in the reference;
every reduction over features runs in ``csrc/repr.hip``.  There is no CPU fallback: CPU tensors raise.
"""
from typing import Dict, List, Optional

import torch

from . import kernels as K
from .hooks import OutputHook


class ReprPredictor:
    """``predict_step`` of the reference's LightningModule for one model.

    ``repr_module_name`` is the hooked module whose output is the representation (``cls_head.avg_pool`` in every CIL
    config, libs/cil/cil.py:432-446)."""

    def __init__(self, model, repr_module_name: str = 'cls_head.avg_pool', extract_repr: bool = True, extract_meta: bool = False):
        self.model = model
        self.repr_module_name = repr_module_name
        self.extract_repr = extract_repr
        self.extract_meta = extract_meta
        self._hook = OutputHook(model, outputs=[repr_module_name], as_tensor=True)

    def close(self):
        self._hook.remove()

    @torch.no_grad()
    def predict_step(self, batch_data: Dict, batch_idx: int = 0) -> Dict:
        x = batch_data['imgs']
        cls_score = self.model(x, return_loss=False)
        result = {'cls_score': cls_score, 'label': batch_data['label']}
        if self.extract_repr:
            feat = self._hook.get_layer_output(self.repr_module_name).flatten(1)        # (B*crops*T, D)
            T = self.model.cls_head.num_segments
            B = x.size(0)
            if feat.size(0) % (B * T) != 0:
                raise ValueError(f'{feat.size(0)} pooled rows are not a multiple of batch {B} x num_segments {T}')
            crops = feat.size(0) // (B * T)
            feat = feat if feat.is_contiguous() else feat.contiguous()
            repr_, mean_crops = K.repr_from_features(feat, B, crops, T)
            result['repr_'] = repr_                               # (batch_size, num_crops, dim)
            result['mean_crops_repr_'] = mean_crops               # (batch_size, dim)
            assert result['repr_'].size(0) == result['cls_score'].size(0)
        if self.extract_meta:
            for k, v in batch_data.items():
                if k not in ['label', 'imgs', 'blended']:
                    result[k] = v
        return result


def nme_classify(repr_: torch.Tensor, exemplar_class_means: torch.Tensor):
    """(num_samples, num_crops, dim) x (num_classes, dim) -> (similarity (num_samples, num_classes), preds_nme)."""
    return K.nme_classify(repr_.contiguous(), exemplar_class_means.contiguous())


def class_means_from_repr(mean_crops_repr: torch.Tensor, label: torch.Tensor, num_classes: int) -> torch.Tensor:
    """``_get_exemplar_class_means`` arithmetic: per-class mean of the crop-averaged representations."""
    r = mean_crops_repr.reshape(-1, mean_crops_repr.size(-1)).contiguous()
    lab = label.reshape(-1).contiguous()
    return K.class_means(r, lab, num_classes)


# Per-sample entries of a prediction dictionary (``predict_step`` output gathered over a class's training videos): tensors indexed
# along dimension 0 by sample, plus the ``frame_dir`` list.  An exemplar entry keeps the second tuple's keys of the chosen samples.
_SAMPLE_TENSOR_KEYS = ('total_frames', 'label', 'clip_len', 'num_clips', 'frame_inds', 'repr_', 'cls_score')
_EXEMPLAR_TENSOR_KEYS = ('total_frames', 'label', 'clip_len', 'frame_inds')
_STORING = {'videos': 3, 'clips': 4}        # storing method -> rank of ``repr_``: (videos, samples, dims) / (videos, clips, samples, dims)


def _take(record: Dict, rows, tensor_keys) -> Dict:
    """Sub-record of the samples ``rows`` (a list of ints or an index tensor)."""
    picked = rows.tolist() if torch.is_tensor(rows) else list(rows)
    out = {'frame_dir': [record['frame_dir'][r] for r in picked]}
    out.update({k: record[k][rows] for k in tensor_keys})
    return out


class Herding:
    """iCaRL herding behind the interface of libs/cil/memory_selection.py:7-164: ``Herding(budget_size, class_indices,
    cosine_distance, storing_methods, budget_type).construct_exemplar(prediction_with_meta)`` returns, per class index,
    ``{'indices', 'dist', 'class_mean', 'frame_dir', 'total_frames', 'label', 'clip_len', 'frame_inds'}`` of the selected samples."""

    def __init__(self, budget_size: int, class_indices: List[int], cosine_distance: bool, storing_methods='clips',
                 budget_type='class'):
        assert storing_methods in ('videos', 'clips', 'frames'), storing_methods
        assert budget_type in ('fixed', 'class'), budget_type
        self.budget_size, self.budget_type = budget_size, budget_type
        self.storing_methods, self.cosine_distance = storing_methods, cosine_distance
        self.class_indices = class_indices
        self.num_classes = len(class_indices)
        # 'class': the budget is per class; 'fixed': one budget shared evenly by the classes seen so far
        self.num_exemplars_per_class = budget_size if budget_type == 'class' else budget_size // self.num_classes

    # -- feature layout handling (memory_selection.py:50-69) --------------------------------------------------------
    def _class_features(self, features: torch.Tensor) -> torch.Tensor:
        if self.storing_methods == 'videos':
            if features.size(1) == 1:
                return features.squeeze(dim=1)
            v, s, d = features.shape                                     # (videos, samples, dims) -> (videos, dims)
            return K.consensus_fwd(features.reshape(v * s, d).contiguous(), v, s)
        if self.storing_methods == 'clips':
            features = features.reshape(-1, features.size(2), features.size(3))
            if features.size(1) == 1:
                return features.squeeze(dim=1)
            # Reference quirk kept for drop-in parity (memory_selection.py:63-69): after the reshape above the tensor is
            # 3-D, so the reference's ``features.mean(2)`` averages over the feature dimension and the herding runs on
            # (videos x clips, samples) vectors.
            vc, s, d = features.shape
            return K.consensus_fwd(features.reshape(vc * s * d, 1).contiguous(), vc * s, d).view(vc, s)
        raise NotImplementedError

    def select(self, features: torch.Tensor):
        """One class: (n, dims) -> (class_mean (1, dims), indices list, dist list)."""
        feats = features.contiguous()
        if self.num_exemplars_per_class > feats.size(0):
            raise ValueError(f'{self.num_exemplars_per_class} exemplars requested from {feats.size(0)} samples')
        cm, idx, dist = K.herding_select(feats, self.num_exemplars_per_class, self.cosine_distance)
        return cm, idx.tolist(), dist.tolist()

    def construct_exemplar(self, prediction_with_meta: Dict) -> Dict:
        self._check_dimension(prediction_with_meta['repr_'], prediction_with_meta['label'])
        meta_by_class = self.split_meta_by_class(prediction_with_meta)
        exemplar_meta = {}
        with torch.no_grad():
            for class_idx, meta in meta_by_class.items():
                features = self._class_features(meta['repr_'])
                class_mean, indices, dist = self.select(features)
                exemplar_meta[class_idx] = {'indices': indices, 'dist': dist, 'class_mean': class_mean}
        return self._update_exemplar(exemplar_meta, meta_by_class)

    def _update_exemplar(self, exemplar_meta: dict, meta_by_class: dict):
        """Attach the bookkeeping of the selected samples (what an exemplar annotation line is written from) to each class entry."""
        for cls, chosen in exemplar_meta.items():
            chosen.update(_take(meta_by_class[cls], chosen['indices'], _EXEMPLAR_TENSOR_KEYS))
        return exemplar_meta

    def _check_dimension(self, all_features, labels):
        if all_features.size(0) != labels.size(0):
            raise ValueError('all_features and labels must have the same value of dim 0')
        want = _STORING.get(self.storing_methods)
        if want is None:                                    # 'frames' passes the constructor, as in the reference, and stops here
            raise NotImplementedError('frame herding not supported yet')
        if all_features.dim() != want:
            layout = '(videos, samples, dims)' if want == 3 else '(videos, clips, samples, dims)'
            raise ValueError(f'Expecting {want}D features: {layout}')

    def split_meta_by_class(self, prediction_with_meta: dict):
        """{class index: the per-sample entries of that class's samples}, in ``class_indices`` order."""
        labels = prediction_with_meta['label']
        return {c: _take(prediction_with_meta, torch.nonzero(labels == c, as_tuple=True)[0], _SAMPLE_TENSOR_KEYS)
                for c in self.class_indices}
