"""Dev tool (CPU): rank clip seeds of the small TSM-R18 parity case by how far the closest pre-ReLU activation of the fp64
oracle stays from zero, and the closest pair of candidates of a stem max-pool window from each other (both in units of the
channel's standard deviation).  tests/test_model_gpu.py uses the best seed for the case in which no ReLU sign and no pool
arg-max is expected to differ between implementations, so that every gradient is held to the strict bar.

    python tools/find_flip_free_seed.py [first_seed] [n_seeds]
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

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
    stem = rec.pre[0]
    act = stem.clamp_min(0)
    N, C, H, W = act.shape
    win = F.unfold(act.reshape(N * C, 1, H, W), 3, padding=1, stride=2)          # (N*C, 9, Ho*Wo), zero padding (act >= 0)
    top2 = win.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).view(N, C, -1) / stem.std(dim=(0, 2, 3)).view(1, C, 1)
    pool_margin = float(gap[top2[:, 0].view(N, C, -1) > 0].min())
    rows.append((min(margin, pool_margin), seed))
    print(f'seed {seed}: min |pre| / channel std = {margin:.3e}, closest pool candidates {pool_margin:.3e}', flush=True)
rows.sort(reverse=True)
print('best:', rows[:5])
