"""Generate golden vectors for the herding selection from the reference's own ``libs/cil/memory_selection.py``.

Run ONCE in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_herding.py

The module is pure torch and is imported by file path; ``Herding.construct_exemplar`` is driven with seeded synthetic
predictions.  Only inputs/outputs (data) are written to ``tests/golden/herding_golden.npz``.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'herding_golden.npz')

# (storing_methods, cosine, num_classes, videos, clips, samples, dims, budget)
CASES = [
    ('videos', True, 3, 40, 0, 1, 64, 5),
    ('videos', False, 3, 40, 0, 1, 64, 5),
    ('videos', True, 2, 30, 0, 3, 128, 8),
    ('clips', True, 3, 24, 2, 1, 96, 6),
    ('clips', False, 2, 20, 2, 2, 32, 4),
    ('videos', True, 2, 50, 0, 1, 2048, 20),     # R50 representation width, UCF101 budget
    ('videos', True, 2, 16, 0, 1, 512, 8),       # whole class selected (budget == class size for one class)
]


def main():
    spec = importlib.util.spec_from_file_location('ref_memory_selection', os.path.join(REF, 'libs/cil/memory_selection.py'))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    out = {}
    for ci, (method, cosine, ncls, videos, clips, samples, dims, budget) in enumerate(CASES):
        g = torch.Generator().manual_seed(4000 + ci)
        if ci == 6:
            labels = torch.tensor([0] * 8 + [1] * 8)
        else:
            labels = torch.randint(0, ncls, (videos,), generator=g)
            labels[:ncls] = torch.arange(ncls)
            while min((labels == c).sum().item() for c in range(ncls)) * max(clips, 1) < budget:
                labels = torch.randint(0, ncls, (videos,), generator=g)
        shape = (videos, samples, dims) if method == 'videos' else (videos, clips, samples, dims)
        feats = torch.randn(shape, generator=g) + 0.5 * torch.randn(1, dims, generator=g)
        if ci == 2:
            feats[5] = feats[3]                      # duplicated sample: exact distance tie, argmin keeps the first
        pred = {'repr_': feats, 'label': labels, 'frame_dir': [f'v{i}' for i in range(videos)],
                'total_frames': torch.arange(videos) + 30, 'clip_len': torch.ones(videos, dtype=torch.long),
                'num_clips': torch.full((videos,), 8), 'frame_inds': torch.arange(videos * 8).view(videos, 8),
                'cls_score': torch.randn(videos, ncls, generator=g)}
        h = ms.Herding(budget_size=budget, class_indices=list(range(ncls)), cosine_distance=cosine, storing_methods=method,
                       budget_type='class')
        if method == 'clips':
            # with clips > 1 the selected indices address (video, clip) rows while the meta lists are per video, so the
            # reference's own _update_exemplar runs out of range; the selection itself is what is pinned here
            h._update_exemplar = lambda exemplar_meta, meta_by_class: exemplar_meta
        ex = h.construct_exemplar(pred)
        out[f'c{ci}_feats'] = feats.numpy()
        out[f'c{ci}_labels'] = labels.numpy()
        out[f'c{ci}_cfg'] = np.array([method == 'clips', cosine, ncls, budget], dtype=np.int64)
        for c in range(ncls):
            out[f'c{ci}_k{c}_indices'] = np.array(ex[c]['indices'], dtype=np.int64)
            out[f'c{ci}_k{c}_dist'] = np.array(ex[c]['dist'], dtype=np.float32)
            out[f'c{ci}_k{c}_class_mean'] = ex[c]['class_mean'].numpy()
            if method == 'videos':
                out[f'c{ci}_k{c}_total_frames'] = ex[c]['total_frames'].numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
