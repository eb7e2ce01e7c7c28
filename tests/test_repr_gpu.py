"""Parity of the representation path (csrc/repr.hip behind bdvcil_amd.representation) against the reference-generated
herding golden vectors and the CPU oracle.  Index outputs must match exactly; floats within 2e-6 (fp32, only the
summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import repr_oracle as R
from oracle import tsm_oracle as O
from test_repr_cpu import iter_golden_classes

pytestmark = pytest.mark.gpu


def test_herding_select_matches_reference_golden(dev):
    from bdvcil_amd import kernels as K
    for g in iter_golden_classes():
        f = R.herding_class_features(g['feats'], g['method']).contiguous().to(dev)
        cm, idx, dist = K.herding_select(f, g['budget'], g['cosine'])
        assert idx.tolist() == g['indices'], (g['case'], g['cls'])
        assert torch.allclose(dist.cpu(), g['dist'], rtol=0, atol=2e-6)
        assert torch.allclose(cm.cpu(), g['class_mean'], rtol=0, atol=2e-7)


def test_herding_class_mirrors_reference_dictionary(dev):
    import bdvcil_amd as bd
    gz = np.load(__import__('test_repr_cpu').GOLD)
    for ci, method in ((0, 'videos'), (1, 'videos'), (2, 'videos'), (3, 'clips')):
        clips, cosine, ncls, budget = [int(v) for v in gz[f'c{ci}_cfg']]
        feats = torch.from_numpy(gz[f'c{ci}_feats']).to(dev)
        labels = torch.from_numpy(gz[f'c{ci}_labels']).to(dev)
        V = feats.shape[0]
        pred = {'repr_': feats, 'label': labels, 'frame_dir': [f'v{i}' for i in range(V)],
                'total_frames': torch.arange(V, device=dev) + 30, 'clip_len': torch.ones(V, dtype=torch.long, device=dev),
                'num_clips': torch.full((V,), 8, device=dev), 'frame_inds': torch.arange(V * 8, device=dev).view(V, 8),
                'cls_score': torch.zeros(V, ncls, device=dev)}
        h = bd.Herding(budget_size=budget, class_indices=list(range(ncls)), cosine_distance=bool(cosine),
                       storing_methods=method, budget_type='class')
        if method == 'clips':
            h._update_exemplar = lambda exemplar_meta, meta_by_class: exemplar_meta     # as in make_golden_herding.py
        ex = h.construct_exemplar(pred)
        for c in range(ncls):
            assert ex[c]['indices'] == gz[f'c{ci}_k{c}_indices'].tolist()
            assert np.allclose(np.array(ex[c]['dist'], dtype=np.float32), gz[f'c{ci}_k{c}_dist'], atol=2e-6)
            assert ex[c]['class_mean'].shape == (1, feats.shape[-1])
            if method == 'videos':
                assert ex[c]['total_frames'].cpu().tolist() == gz[f'c{ci}_k{c}_total_frames'].tolist()
                assert ex[c]['frame_dir'] == [f'v{i}' for i in (labels == c).nonzero(as_tuple=True)[0][ex[c]['indices']].tolist()]
    with pytest.raises(ValueError):
        bd.Herding(budget_size=100, class_indices=[0], cosine_distance=True, storing_methods='videos').select(feats.reshape(feats.shape[0], -1)[:5])
    with pytest.raises(ValueError):
        bd.Herding(4, [0], True, 'videos').construct_exemplar({'repr_': feats.reshape(feats.shape[0], -1), 'label': labels})   # 2-D features


def test_nme_and_class_means_vs_oracle(dev):
    import bdvcil_amd as bd
    g = torch.Generator().manual_seed(3)
    for (S, crops, D, K) in [(7, 1, 64, 5), (33, 10, 2048, 101), (5, 3, 512, 51)]:
        r = F.normalize(torch.randn(S, crops, D, generator=g), dim=-1)
        means = torch.randn(K, D, generator=g)
        means[K // 2] = means[0]                                   # duplicated class mean: arg-max tie -> first index
        sim_ref, pred_ref = R.nme_classify(r, means)
        sim, pred = bd.nme_classify(r.to(dev), means.to(dev))
        assert torch.allclose(sim.cpu(), sim_ref, rtol=0, atol=2e-6)
        assert torch.equal(pred.cpu(), pred_ref)
        labels = torch.randint(0, K, (S,), generator=g)
        cm_ref = R.class_means(r.mean(1), labels, K)
        cm = bd.class_means_from_repr(r.mean(1).to(dev), labels.to(dev), K).cpu()
        assert torch.equal(torch.isnan(cm), torch.isnan(cm_ref))
        assert torch.allclose(torch.nan_to_num(cm), torch.nan_to_num(cm_ref), rtol=0, atol=1e-6)


def test_repr_kernel_vs_oracle(dev):
    from bdvcil_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    for (B, crops, T, D) in [(2, 1, 8, 512), (3, 10, 8, 2048), (1, 3, 8, 64)]:
        pooled = torch.randn(B * crops * T, D, generator=g).abs()      # pooled ReLU features are non-negative
        r_ref, m_ref = R.predict_repr(pooled, B, T)
        r, m = K.repr_from_features(pooled.to(dev), B, crops, T)
        assert torch.allclose(r.cpu(), r_ref, rtol=0, atol=2e-7)
        assert torch.allclose(m.cpu(), m_ref, rtol=0, atol=2e-7)
    with pytest.raises(ValueError):
        K.repr_from_features(pooled.to(dev), 2, 3, 8)


def test_predict_step_end_to_end_r18(dev):
    """TenCrop-style test clips through the eval forward: cls_score, repr_, mean_crops_repr_ vs the CPU oracle."""
    import bdvcil_amd as bd
    torch.manual_seed(0)
    cfg = O.r50_cfg(num_classes=11, depth=18, head='LocalSimilarityClassifier', loss='LSCLoss', dropout_ratio=0.5)
    om = O.build_model(cfg).eval()
    hm = bd.build_model(cfg).to(dev).eval()
    hm.load_state_dict(om.state_dict())
    B, crops, T = 2, 3, 8
    g = torch.Generator().manual_seed(7)
    imgs = torch.randn(B, crops * T, 3, 64, 64, generator=g)
    label = torch.randint(0, 11, (B, 1), generator=g)
    tap = O.FeatureTap(om, ['cls_head.avg_pool'])
    with torch.no_grad():
        score_ref = om(imgs, return_loss=False)
    r_ref, m_ref = R.predict_repr(tap.out['cls_head.avg_pool'], B, T)
    pr = bd.ReprPredictor(hm, extract_meta=True)
    res = pr.predict_step({'imgs': imgs.to(dev), 'label': label.to(dev), 'frame_dir': ['a', 'b'], 'blended': None})
    pr.close()
    assert set(res) == {'cls_score', 'label', 'repr_', 'mean_crops_repr_', 'frame_dir'}
    assert res['repr_'].shape == (B, crops, 512) and res['mean_crops_repr_'].shape == (B, 512)
    assert torch.allclose(res['cls_score'].cpu(), score_ref, rtol=0, atol=1e-3)
    assert torch.equal(res['cls_score'].cpu().argmax(1), score_ref.argmax(1))
    assert torch.allclose(res['repr_'].cpu(), r_ref, rtol=0, atol=1e-4)
    assert torch.allclose(res['mean_crops_repr_'].cpu(), m_ref, rtol=0, atol=1e-4)
