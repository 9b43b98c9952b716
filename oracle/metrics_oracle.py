"""TEST INFRASTRUCTURE ONLY - CPU restatement (torch, float64-capable) of the contrastive monitoring metrics of the
reference; pinned by tests/golden/metrics.npz, which tests/golden/gen_golden_metrics.py generates by
importing the reference's own trainer/metrics.py.  Only tests/ may import this module."""
import torch


def _cos_matrix(x1, x2):
    """trainer/metrics.py:244-248: S / (|x1_i| |x2_j|), x2 cut to len(x1) rows (:241-242)."""
    x2 = x2[:x1.shape[0]]
    return (x1 @ x2.T) / torch.outer(x1.norm(dim=1), x2.norm(dim=1))


def positive_similarity(x1, x2):                      # :310-333 (global-global branch)
    return float(((torch.diagonal(_cos_matrix(x1, x2)) + 1) / 2).mean())


def negative_similarity(x1, x2):                      # :443-463
    c = _cos_matrix(x1, x2)
    B = x1.shape[0]
    return float((((c.sum(1) - torch.diagonal(c)) / (B - 1) + 1) / 2).mean())


def rates(x1, x2, threshold):                         # :237-308: a pair counts as "predicted positive" if (cos+1)/2 > threshold
    c = _cos_matrix(x1, x2)
    B = x1.shape[0]
    pred = (c + 1) / 2 > threshold
    eye = torch.eye(B, dtype=torch.bool)
    tpr = float(pred[eye].sum()) / B
    tnr = float((~pred[~eye]).sum()) / (B * (B - 1))
    return tpr, tnr, (tpr + tnr) / 2


def uniformity(x1, x2, t=2):                          # commons/losses.py:946-951
    def one(x):
        d2 = torch.pdist(x, p=2).pow(2)
        return torch.log(torch.exp(-t * d2).mean())
    return float((one(x1) + one(x2)) / 2)


def alignment(x1, x2, alpha=2):                       # :216-225
    return float((x1 - x2[:x1.shape[0]]).norm(dim=1).pow(alpha).mean())


def batch_variance(x1, x2):                           # :169-174
    return float(x1.std(dim=0).mean() + x2.std(dim=0).mean())


def dimension_covariance(x1, x2):                     # :161-166, commons/losses.py:954-959
    def one(x):
        n, d = x.shape
        xc = x - x.mean(0)
        cov = xc.T @ xc / (n - 1)
        off = cov - torch.diag(torch.diagonal(cov))
        return (off ** 2).sum() / d
    return float(one(x1) + one(x2))


def all_metrics(x1, x2, threshold=0.5009, t=2, alpha=2):
    tpr, tnr, acc = rates(x1, x2, threshold)
    return {'positive_similarity': positive_similarity(x1, x2), 'negative_similarity': negative_similarity(x1, x2),
            'true_positive_rate': tpr, 'true_negative_rate': tnr, 'contrastive_accuracy': acc,
            'uniformity': uniformity(x1, x2, t), 'alignment': alignment(x1, x2, alpha),
            'batch_variance': batch_variance(x1, x2), 'dimension_covariance': dimension_covariance(x1, x2),
            'mean_pred': float(x1.mean()), 'std_pred': float(x1.std()), 'mean_targets': float(x2.mean()),
            'std_targets': float(x2.std())}
