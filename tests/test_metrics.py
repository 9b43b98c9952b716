"""the contrastive monitoring metrics.  CPU: the oracle against the golden values generated from the
reference's own trainer/metrics.py.  GPU: the one-pass HIP implementation against the same golden values and the oracle."""
import importlib

import numpy as np
import pytest
import torch

from helpers import load
from oracle import metrics_oracle as MO

NAMES = ['positive_similarity', 'negative_similarity', 'contrastive_accuracy', 'true_negative_rate', 'true_positive_rate',
         'uniformity', 'alignment', 'batch_variance', 'dimension_covariance', 'mean_pred', 'std_pred', 'mean_targets',
         'std_targets']
CASES = ['b8', 'b64', 'b500', 'noisy']


def _close(a, b, rel=2e-5, floor=1e-7):
    if a == b:           # also -inf == -inf (uniformity of far-apart embeddings underflows, in the reference too)
        return True
    return abs(a - b) <= rel * max(abs(a), abs(b)) + floor


@pytest.mark.parametrize('case', CASES)
def test_oracle_metrics_match_reference_values(case):
    z = load('metrics.npz')
    z1, z2 = torch.from_numpy(z[f'{case}/z1']), torch.from_numpy(z[f'{case}/z2'])
    got = MO.all_metrics(z1, z2) if case != 'noisy' else None
    for name in NAMES:
        key = f'{case}/{name}'
        if key not in z.files:
            continue
        if got is None:      # noisy case: the metrics that the reference defines on the cut x2
            val = {'positive_similarity': MO.positive_similarity, 'negative_similarity': MO.negative_similarity,
                   'alignment': MO.alignment}.get(name)
            if val is not None:
                v = val(z1, z2)
            elif name in ('true_positive_rate', 'true_negative_rate', 'contrastive_accuracy'):
                tpr, tnr, acc = MO.rates(z1, z2, 0.5009)
                v = {'true_positive_rate': tpr, 'true_negative_rate': tnr, 'contrastive_accuracy': acc}[name]
            else:
                v = {'mean_pred': float(z1.mean()), 'std_pred': float(z1.std()), 'mean_targets': float(z2.mean()),
                     'std_targets': float(z2.std())}[name]
        else:
            v = got[name]
        assert _close(v, float(z[key])), (key, v, float(z[key]))


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_hip_metrics_match_reference_values(case):
    M = importlib.import_module('3dinfomax_amd.metrics')
    z = load('metrics.npz')
    z1 = torch.from_numpy(z[f'{case}/z1']).cuda()
    z2 = torch.from_numpy(z[f'{case}/z2']).cuda()
    objs = {'positive_similarity': M.PositiveSimilarity(), 'negative_similarity': M.NegativeSimilarity(),
            'contrastive_accuracy': M.ContrastiveAccuracy(threshold=0.5009), 'true_negative_rate': M.TrueNegativeRate(threshold=0.5009),
            'true_positive_rate': M.TruePositiveRate(threshold=0.5009), 'uniformity': M.Uniformity(t=2),
            'alignment': M.Alignment(alpha=2), 'batch_variance': M.BatchVariance(), 'dimension_covariance': M.DimensionCovariance()}
    for name, m in objs.items():
        key = f'{case}/{name}'
        if key not in z.files:
            continue
        v = m(z1, z2).item()                     # the trainer's call pattern (trainer/self_supervised_trainer.py:49)
        assert _close(v, float(z[key]), rel=2e-4, floor=2e-6), (key, v, float(z[key]))
    allv = M.contrastive_metrics(z1, z2)
    for name in ('mean_pred', 'std_pred', 'mean_targets', 'std_targets'):
        assert _close(allv[name], float(z[f'{case}/{name}']), rel=2e-4, floor=2e-6), name
    with pytest.raises(NotImplementedError):
        objs['positive_similarity'](z1, z2, pos_mask=torch.eye(len(z1), device='cuda'))


@pytest.mark.gpu
def test_hip_metrics_share_one_pass_and_track_tensor_versions():
    M = importlib.import_module('3dinfomax_amd.metrics')
    g = torch.Generator().manual_seed(5)
    z1, z2 = (torch.randn(96, 64, generator=g) * 0.2).cuda(), (torch.randn(96, 64, generator=g) * 0.2).cuda()
    a = M.PositiveSimilarity()(z1, z2).item()
    vals = M._cache['values']
    assert M.NegativeSimilarity()(z1, z2).item() == pytest.approx(vals['negative_similarity'])
    assert M._cache['values'] is vals                       # second metric: cached, no new pass
    z1.mul_(-1.0)                                           # in-place update bumps the version -> recomputed
    b = M.PositiveSimilarity()(z1, z2).item()
    assert M._cache['values'] is not vals and abs((a - 0.5) + (b - 0.5)) < 1e-5
    ref = MO.all_metrics(z1.cpu(), z2.cpu())
    for k, v in M.contrastive_metrics(z1, z2).items():
        assert _close(v, ref[k], rel=2e-4, floor=2e-6), k
