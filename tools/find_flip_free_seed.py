"""Dev tool (CPU): rank clip seeds of the small TSM-R18 parity case by how far the closest pre-ReLU activation of the fp64
oracle stays from zero (in units of its channel's standard deviation).  tests/test_model_gpu.py uses the best seed for the
case in which no ReLU sign is expected to differ between implementations, so that every gradient is held to the strict bar.

    python tools/find_flip_free_seed.py [first_seed] [n_seeds]
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import tsm_oracle as O
from tests.test_model_gpu import ReluRecorder, _clips, _oracle_only

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ref = _oracle_only(18, 'LocalSimilarityClassifier', 'LSCLoss', K=11).double()
ref.train()
state = copy.deepcopy(ref.state_dict())
rows = []
for seed in range(first, first + n):
    ref.load_state_dict(state)
    imgs, labels = _clips(2, 8, 64, 11, seed=seed)
    with ReluRecorder() as rec, torch.no_grad():
        ref(imgs.double(), labels)
    margin = min(float((pre.abs() / pre.std(dim=(0, 2, 3), keepdim=True)).min()) for pre in rec.pre)
    rows.append((margin, seed))
    print(f'seed {seed}: min |pre| / channel std = {margin:.3e}', flush=True)
rows.sort(reverse=True)
print('best:', rows[:5])
