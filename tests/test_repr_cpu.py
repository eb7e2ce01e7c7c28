"""CPU tests of the representation-path oracle (oracle/repr_oracle.py): the herding restatement is pinned against
golden vectors produced by the reference's own Herding.construct_exemplar (tests/golden/make_golden_herding.py);
the cil.py snippets (predict_step representation, NME, class means) are checked at definition level."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import repr_oracle as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'herding_golden.npz')
N_CASES = 7


def _t(a):
    return torch.from_numpy(np.asarray(a))


def iter_golden_classes():
    gz = np.load(GOLD)
    for ci in range(N_CASES):
        clips, cosine, ncls, budget = [int(v) for v in gz[f'c{ci}_cfg']]
        feats, labels = _t(gz[f'c{ci}_feats']), _t(gz[f'c{ci}_labels'])
        for c in range(ncls):
            sel = feats[(labels == c).nonzero(as_tuple=True)[0]]
            yield dict(case=ci, cls=c, method='clips' if clips else 'videos', cosine=bool(cosine), budget=budget, feats=sel,
                       indices=gz[f'c{ci}_k{c}_indices'].tolist(), dist=_t(gz[f'c{ci}_k{c}_dist']),
                       class_mean=_t(gz[f'c{ci}_k{c}_class_mean']))


def test_herding_oracle_matches_reference_golden():
    n = 0
    for g in iter_golden_classes():
        f = R.herding_class_features(g['feats'], g['method'])
        cm, idx, dist = R.herding_select(f, g['budget'], g['cosine'])
        assert idx == g['indices'], (g['case'], g['cls'])
        assert torch.allclose(torch.tensor(dist), g['dist'], rtol=0, atol=1e-6)
        assert torch.allclose(cm, g['class_mean'], rtol=0, atol=1e-7)
        assert len(set(idx)) == len(idx) == g['budget']
        n += 1
    assert n == 17


def test_golden_has_an_exact_tie_and_a_full_class():
    cases = list(iter_golden_classes())
    full = [g for g in cases if g['case'] == 6]
    assert all(sorted(g['indices']) == list(range(8)) for g in full)        # every sample of the class gets picked
    assert any(g['case'] == 2 for g in cases)


def test_predict_repr_definition():
    g = torch.Generator().manual_seed(0)
    B, crops, T, D = 3, 2, 8, 32
    pooled = torch.randn(B * crops * T, D, 1, 1, generator=g)
    r, m = R.predict_repr(pooled, B, T)
    assert r.shape == (B, crops, D) and m.shape == (B, D)
    ref = pooled.flatten(1).view(B, crops, T, D).mean(2)
    ref = ref / ref.norm(dim=-1, keepdim=True)
    assert torch.allclose(r, ref, atol=1e-6)
    assert torch.allclose(m, ref.mean(1), atol=1e-6)


def test_nme_and_class_means_definition():
    g = torch.Generator().manual_seed(1)
    S, crops, D, K = 9, 3, 16, 5
    r = F.normalize(torch.randn(S, crops, D, generator=g), dim=-1)
    means = torch.randn(K, D, generator=g)
    sim, pred = R.nme_classify(r, means)
    ref = torch.einsum('scd,kd->sck', r, F.normalize(means, dim=-1)).mean(1)
    assert torch.allclose(sim, ref, atol=1e-6) and torch.equal(pred, ref.argmax(1))
    labels = torch.tensor([0, 1, 2, 0, 1, 2, 4, 4, 0])
    cm = R.class_means(r.mean(1), labels, K)
    assert torch.allclose(cm[4], r.mean(1)[6:8].mean(0), atol=1e-7)
    assert torch.isnan(cm[3]).all()                                          # empty class: mean of nothing
