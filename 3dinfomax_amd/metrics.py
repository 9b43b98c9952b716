"""Contrastive monitoring metrics - drop-in for the classes of reference trainer/metrics.py that
configs_clean/pre-train_QM9.yml:15-24 lists.

The reference trainer calls every metric separately and `.item()`s each result, every `log_iterations` (= 2) steps
(trainer/self_supervised_trainer.py:31-50, train.py:255-268): nine chains of small device ops and nine host syncs.
Here the first metric asked about a pair of embedding tensors computes ALL of them in one device pass (five small GEMMs
+ two row kernels + column sums, csrc/metrics.hip) and brings them to the host with one copy; the other metric objects
find the result cached, so the trainer's loop costs one synchronisation.  Same class names, constructor arguments and
call signature; the value comes back as a 0-dim CPU tensor (`.item()` works as in the trainer).  The `pos_mask`
(local-global losses) variants are outside the scope of this path and raise NotImplementedError.
"""
import math
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops

# one entry: (weakref(x1), weakref(x2), key, values).  The key alone (address, version, shape) is NOT an identity: each
# training step's detached embeddings are new tensors with version 0, the same shape and - through the caching allocator -
# very likely the same address, so the cache also requires that the very tensor OBJECTS are the ones it was filled from
# (weak references: a dead tensor can never match, and the cache keeps no embedding alive).
_cache = {'key': None, 'values': None, 'x1': None, 'x2': None}


def _same_object(ref, t):
    return ref is not None and ref() is t


def _log(v):
    with np.errstate(divide='ignore'):       # exp(-t d^2) underflows to 0 for far-apart embeddings: log -> -inf, as torch.log
        return float(np.log(v))


def _all_metrics(x1, x2, threshold=0.5, t=2.0, alpha=2.0):
    key = (x1.data_ptr(), x1._version, tuple(x1.shape), x2.data_ptr(), x2._version, tuple(x2.shape), float(threshold), float(t),
           float(alpha))
    if _cache['key'] == key and _same_object(_cache['x1'], x1) and _same_object(_cache['x2'], x2):
        return _cache['values']
    with torch.no_grad():
        z1, z2 = x1.detach().float().contiguous(), x2.detach().float().contiguous()
        if not z1.is_cuda:
            raise AssertionError('the metrics run on the HIP device (there is no CPU fallback)')
        B1, D = z1.shape
        B2 = z2.shape[0]
        if B2 < B1 or z2.shape[1] != D:
            raise ValueError(f'incompatible embedding shapes {tuple(z1.shape)} / {tuple(z2.shape)}')
        L = _lib.load()
        S = ops.gemm(z1, z2[:B1], trans_b=True)            # x2 may carry extra "noisy" rows at the end (metrics.py:222)
        G1 = ops.gemm(z1, z1, trans_b=True)
        G2 = ops.gemm(z2, z2, trans_b=True)
        rows = torch.empty(B2, 8, dtype=torch.float32, device=z1.device)
        _lib.check(L.i3d_contrastive_rowstats(S.data_ptr(), G1.data_ptr(), G2.data_ptr(), B1, B2, float(threshold), float(t),
                                              float(alpha), rows.data_ptr(), ops._stream()), 'i3d_contrastive_rowstats')
        parts = [ops.colsum(rows)]
        for z, n in ((z1, B1), (z2, B2)):
            C = ops.gemm(z, z, trans_a=True)               # [D, D] second moments
            s = ops.colsum(z)
            cr = torch.empty(D, 2, dtype=torch.float32, device=z.device)
            _lib.check(L.i3d_cov_rowstats(C.data_ptr(), s.data_ptr(), n, D, cr.data_ptr(), ops._stream()), 'i3d_cov_rowstats')
            parts += [ops.colsum(cr)[:1], s, cr[:, 1].contiguous()]
        host = torch.cat([p.reshape(-1) for p in parts]).cpu().double().numpy()     # the one device-to-host copy
    r = host[:8]
    o = 8
    cov1, s1, q1 = host[o], host[o + 1:o + 1 + D], host[o + 1 + D:o + 1 + 2 * D]
    o += 1 + 2 * D
    cov2, s2, q2 = host[o], host[o + 1:o + 1 + D], host[o + 1 + D:o + 1 + 2 * D]

    def col_std_mean(s, q, n):
        return float(np.sqrt(np.maximum(q - s * s / n, 0.0) / max(n - 1, 1)).mean())

    def all_mean_std(s, q, n):
        cnt = n * D
        mean = s.sum() / cnt
        return float(mean), float(math.sqrt(max(q.sum() - s.sum() ** 2 / cnt, 0.0) / max(cnt - 1, 1)))

    tpr = r[2] / B1
    tnr = r[3] / (B1 * (B1 - 1)) if B1 > 1 else float('nan')
    vals = {
        'positive_similarity': (r[1] / B1 + 1) / 2,
        'negative_similarity': ((r[0] - r[1]) / (B1 * (B1 - 1)) + 1) / 2 if B1 > 1 else float('nan'),
        'true_positive_rate': tpr, 'true_negative_rate': tnr, 'contrastive_accuracy': (tpr + tnr) / 2,
        'alignment': r[4] / B1,
        'uniformity': (_log(r[5] / (B1 * (B1 - 1) / 2)) + _log(r[6] / (B2 * (B2 - 1) / 2))) / 2 if B1 > 1 else float('nan'),
        'batch_variance': col_std_mean(s1, q1, B1) + col_std_mean(s2, q2, B2),
        'dimension_covariance': cov1 / D + cov2 / D,
    }
    vals['mean_pred'], vals['std_pred'] = all_mean_std(s1, q1, B1)
    vals['mean_targets'], vals['std_targets'] = all_mean_std(s2, q2, B2)
    _cache['key'], _cache['values'] = key, vals
    _cache['x1'], _cache['x2'] = weakref.ref(x1), weakref.ref(x2)
    return vals


def contrastive_metrics(z2d, z3d, threshold=0.5009, t=2.0, alpha=2.0):
    """All metrics (and the four statistics of SelfSupervisedTrainer.evaluate_metrics, :33-36) as a dict of floats."""
    return dict(_all_metrics(z2d, z3d, threshold, t, alpha))


class _Metric(nn.Module):
    name = None

    def _kw(self):
        return {}

    def forward(self, x1, x2, pos_mask=None):
        if pos_mask is not None:
            raise NotImplementedError('pos_mask (local-global contrastive losses) is outside the accelerated path')
        return torch.tensor(_all_metrics(x1, x2, **{**_shared_defaults(), **self._kw()})[self.name], dtype=torch.float32)


_defaults = {'threshold': 0.5009, 't': 2.0, 'alpha': 2.0}        # train.py:261-266: the values the reference constructs with


def _shared_defaults():
    return dict(_defaults)


class PositiveSimilarity(_Metric):
    """reference trainer/metrics.py:310-333."""
    name = 'positive_similarity'


class NegativeSimilarity(_Metric):
    """reference trainer/metrics.py:443-463."""
    name = 'negative_similarity'


class _Thresholded(_Metric):
    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    def _kw(self):
        return {'threshold': self.threshold}


class TruePositiveRate(_Thresholded):
    """reference trainer/metrics.py:237-258."""
    name = 'true_positive_rate'


class TrueNegativeRate(_Thresholded):
    """reference trainer/metrics.py:261-283."""
    name = 'true_negative_rate'


class ContrastiveAccuracy(_Thresholded):
    """reference trainer/metrics.py:286-310."""
    name = 'contrastive_accuracy'


class Uniformity(_Metric):
    """reference trainer/metrics.py:228-234, commons/losses.py:946-951."""
    name = 'uniformity'

    def __init__(self, t=2):
        super().__init__()
        self.t = t

    def _kw(self):
        return {'t': self.t}


class Alignment(_Metric):
    """reference trainer/metrics.py:216-225."""
    name = 'alignment'

    def __init__(self, alpha=2):
        super().__init__()
        self.alpha = alpha

    def _kw(self):
        return {'alpha': self.alpha}


class BatchVariance(_Metric):
    """reference trainer/metrics.py:169-174."""
    name = 'batch_variance'


class DimensionCovariance(_Metric):
    """reference trainer/metrics.py:161-166, commons/losses.py:954-959."""
    name = 'dimension_covariance'
