"""CPU oracle for the inference + representation path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in plain torch on the CPU, the reference arithmetic that ``csrc/repr.hip`` implements:

* ``predict_repr``        libs/cil/cil.py:501-506 (_extract_repr) and :564-571 (predict_step, extract_repr)
* ``nme_classify``        libs/cil/cil.py:945-960
* ``class_means``         libs/cil/cil.py:1079-1083
* ``herding_select``      libs/cil/memory_selection.py:70-92 (greedy loop) and :150-164 (calc_mean_features)

Pinning: ``herding_select`` is checked against ``tests/golden/herding_golden.npz``, produced by the reference's own
``Herding.construct_exemplar`` (tests/golden/make_golden_herding.py).  The three cil.py snippets live inside a
LightningModule that cannot be imported here (pytorch_lightning / mmaction absent): they are a handful of torch calls
(``F.normalize``, ``F.cosine_similarity``, ``mean``, ``argmax``) restated verbatim -- parity unpinned beyond torch itself.
"""
from typing import Tuple

import torch
import torch.nn.functional as F


def predict_repr(pooled: torch.Tensor, batch_size: int, num_segments: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pooled: hooked ``cls_head.avg_pool`` output (B*crops*T, D, 1, 1) or (B*crops*T, D)."""
    repr_ = pooled.flatten(1)
    repr_ = repr_.view(-1, num_segments, repr_.size(1))
    repr_consensus = repr_.mean(dim=1, keepdim=True).squeeze(1)          # AvgConsensus(dim=1)
    embedding_size = repr_consensus.size(-1)
    r = repr_consensus.view(batch_size, -1, embedding_size)             # (batch_size, num_crops, dim)
    r = F.normalize(r, p=2, dim=-1)
    return r, torch.mean(r, dim=1, keepdim=False)


def nme_classify(repr_: torch.Tensor, exemplar_class_means: torch.Tensor):
    num_samples, num_crops, _ = repr_.shape
    r = repr_.reshape(-1, repr_.size(2))
    num_classes = exemplar_class_means.size(0)
    size, dims = r.shape
    repr_broadcast = r.unsqueeze(dim=1).expand(size, num_classes, dims)
    similarity = F.cosine_similarity(repr_broadcast, exemplar_class_means, dim=-1)
    similarity = torch.mean(similarity.reshape(num_samples, num_crops, num_classes), dim=1, keepdim=False)
    return similarity, torch.argmax(similarity, dim=1, keepdim=False)


def class_means(mean_crops_repr: torch.Tensor, label: torch.Tensor, num_classes: int) -> torch.Tensor:
    r = mean_crops_repr.reshape(-1, mean_crops_repr.size(-1))
    out = []
    for class_idx in range(num_classes):
        indices = (label == class_idx).nonzero().squeeze(dim=1)
        out.append(torch.mean(r[indices], dim=0))
    return torch.stack(out, dim=0)


def _remove_row(t: torch.Tensor, idx) -> torch.Tensor:
    keep = torch.ones(t.shape[0], dtype=torch.bool)
    keep[idx] = False
    return t[keep]


def herding_select(features: torch.Tensor, num_exemplars: int, cosine_distance: bool):
    """features (n, dims) of one class -> (class_mean (1, dims), indices, dist)."""
    if cosine_distance:
        normalized = F.normalize(features, p=2, dim=-1)
    else:
        normalized = features
    mean = features.view(-1, features.size(-1)).mean(0, keepdim=True)
    class_mean = F.normalize(mean, p=2) if cosine_distance else mean
    indexer = torch.arange(features.size(0))
    moving = torch.zeros(1, features.size(-1))
    indices, dists = [], []
    for n in range(1, num_exemplars + 1):
        tmp = moving * (n - 1) / n + normalized / n
        if cosine_distance:
            dist = 1 - torch.cosine_similarity(tmp, class_mean, dim=1)
        else:
            dist = torch.pairwise_distance(tmp, class_mean.squeeze(dim=0), p=2)
        row = torch.argmin(dist)
        moving = moving * (n - 1) / n + normalized[row] / n
        indices.append(indexer[row].item())
        dists.append(dist[row].item())
        normalized = _remove_row(normalized, row)
        indexer = _remove_row(indexer, row)
    return class_mean, indices, dists


def herding_class_features(features: torch.Tensor, storing_methods: str) -> torch.Tensor:
    """memory_selection.py:50-69: per-class feature tensor -> (n, dims)."""
    if storing_methods == 'videos':
        return features.squeeze(dim=1) if features.size(1) == 1 else features.mean(1)
    if storing_methods == 'clips':
        features = features.view(-1, features.size(2), features.size(3))
        if features.size(1) == 1:
            return features.squeeze(dim=1)
        # Reference quirk (:63-69): the tensor is already 3-D (videos x clips, samples, dims) here, so ``mean(2)``
        # averages over the feature dimension and herding then runs on (videos x clips, samples) vectors.
        features = features.mean(2)
        return features.view(-1, features.size(-1))
    raise NotImplementedError
