"""Evaluation metric of the eval path with upstream's interface, computed on the GPU in one pass per batch.

RDFGANMetric <- RDFC-GAN/lib/metrics/rdf_gan_metric.py:6-163 (evaluate_all: per-image RMSE / MAE / iRMSE / iMAE / REL /
delta^1..3 over gt > t_valid, mean over images); evaluate_normalised() additionally folds in Eval.inference's
de-normalisation (lib/evaluator/evaluator.py:17-46) so that the network output can be scored without leaving the device.
"""
import torch

from . import hip


class RDFGANMetric:
    def __init__(self, t_valid=1e-4):
        self.t_valid = t_valid
        self.metric_name = ['RMSE', 'MAE', 'iRMSE', 'iMAE', 'REL', 'D^1', 'D^2', 'D^3']

    def evaluate_normalised(self, gt_norm, pred_norm, evaluate_mask=None, std=5.0, mean=5.0):
        """normalised device maps (N,1,H,W) -> per-image metrics tensor [N,8] (stays on the device)"""
        return hip.depth_metrics(gt_norm, pred_norm, evaluate_mask, std, mean, self.t_valid)

    def evaluate_all(self, results, logger=None, device=None):
        """upstream's signature: `results` is Eval.inference's list of dict(gt=, pd=[, evaluate_mask=]) in metres"""
        device = device or torch.device('cuda', torch.cuda.current_device())
        per = []
        for r in results:
            gt = torch.as_tensor(r['gt']).to(device).float().reshape(1, -1)
            pd = torch.as_tensor(r['pd']).to(device).float().reshape(1, -1)
            ev = r.get('evaluate_mask')
            if ev is not None:
                ev = torch.as_tensor(ev).to(device).reshape(1, -1)
            per.append(hip.depth_metrics(gt, pd, ev, 1.0, 0.0, self.t_valid))       # already in metres: std 1, mean 0
        mean = torch.cat(per).double().mean(dim=0).tolist()
        ret = dict(zip(self.metric_name, mean))
        for k, v in ret.items():
            (logger.log if logger is not None else print)(f'{k}: {v}')
        return ret
