"""ORACLE (test infrastructure): CPU restatement of the evaluation metric of the eval path.

RDFGANMetric.evaluate_all <- RDFC-GAN/lib/metrics/rdf_gan_metric.py:59-151 (per-image RMSE, MAE, iRMSE, iMAE, REL,
delta^1..3 over the pixels with gt > t_valid (and evaluate_mask), then the mean over images); the de-normalisation in
front of it <- RDFC-GAN/lib/evaluator/evaluator.py:17-46 (x * std + mean, std = mean = 5 for NYUv2 depth).
Pinned by tests/golden/metric.npz (generated from the imported reference by tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import numpy as np

NAMES = ('RMSE', 'MAE', 'iRMSE', 'iMAE', 'REL', 'D^1', 'D^2', 'D^3')


def denormalise(x, std=5.0, mean=5.0):
    """evaluator.py:28-29"""
    return np.asarray(x, dtype=np.float32) * np.float32(std) + np.float32(mean)


def image_metrics(gt, pred, evaluate_mask=None, t_valid=1e-4, acc=np.float32):
    """one image (any shape), f32 arithmetic like the reference -> the eight metrics of rdf_gan_metric.py:72-137.
    acc: dtype of the sums (the reference sums in f32; np.float64 gives the value the f32 sums approximate -- iRMSE is a
    sum of squares of numbers up to 1e4 and moves by ~1e-3 relative with the summation order on a full-size image)"""
    gt = np.asarray(gt, dtype=np.float32).reshape(-1)
    pred = np.asarray(pred, dtype=np.float32).reshape(-1)
    pred_inv = np.float32(1.0) / (pred + np.float32(1e-8))           # :74-75
    gt_inv = np.float32(1.0) / (gt + np.float32(1e-8))
    mask = gt > np.float32(t_valid)                                   # :78
    if evaluate_mask is not None:
        mask &= np.asarray(evaluate_mask, dtype=bool).reshape(-1)
    n = acc(mask.sum()) + acc(1e-8)
    pred, gt, pred_inv, gt_inv = pred[mask], gt[mask], pred_inv[mask], gt_inv[mask]
    pred_inv[pred <= np.float32(t_valid)] = 0.0                       # :87-88
    gt_inv[gt <= np.float32(t_valid)] = 0.0
    diff = pred - gt
    rmse = np.sqrt(acc((diff * diff).sum(dtype=acc)) / n)
    mae = acc(np.abs(diff).sum(dtype=acc)) / n
    dinv = pred_inv - gt_inv
    irmse = np.sqrt(acc((dinv * dinv).sum(dtype=acc)) / n)
    imae = acc(np.abs(dinv).sum(dtype=acc)) / n
    rel = acc((np.abs(diff) / (gt + np.float32(1e-8))).sum(dtype=acc)) / n
    ratio = np.maximum(gt / (pred + np.float32(1e-8)), pred / (gt + np.float32(1e-8)))
    d = [acc((ratio < np.float32(1.25 ** k)).sum()) / n for k in (1, 2, 3)]
    return np.array([rmse, mae, irmse, imae, rel] + d, dtype=np.float64)


def evaluate_all(results, t_valid=1e-4):
    """results: list of dict(gt=..., pd=..., [evaluate_mask=...]) -> dict name -> mean over images (:139-146)"""
    per = np.stack([image_metrics(r['gt'], r['pd'], r.get('evaluate_mask'), t_valid) for r in results])
    mean = per.mean(axis=0)
    return {n: float(v) for n, v in zip(NAMES, mean)}, per
