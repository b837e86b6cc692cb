"""Oracle losses.  TEST INFRASTRUCTURE ONLY.

Restates RDFC-GAN/lib/losses/gan_loss.py: L1_loss :6-22, L2_loss :25-42,
mse_loss :104-120, GANLoss :169-206 (lsgan / vanilla / wgan modes).
All reductions are weighted sums; the default weight is 1/(numel + 1e-6).
"""
import torch
import torch.nn.functional as F


def _weighted_sum(elem, pred, weight):
    if weight is None:
        n = torch.ones_like(pred).sum()
        weight = torch.ones_like(pred) / (n + 1e-6)
    else:
        weight = weight.float()
        if weight.dim() != elem.dim():
            weight = weight.unsqueeze(1)
    return (weight * elem).sum()


def l1_loss(pred, target, weight=None):
    return _weighted_sum((pred - target).abs(), pred, weight)


def l2_loss(pred, target, weight=None):
    return _weighted_sum((pred - target) ** 2, pred, weight)


class GANLoss:
    def __init__(self, gan_mode='lsgan', real=1.0, fake=0.0):
        assert gan_mode in ('lsgan', 'vanilla', 'wgan', 'wgangp')
        self.mode, self.real, self.fake = gan_mode, real, fake

    def __call__(self, pred, target_is_real, weight=None):
        if self.mode in ('wgan', 'wgangp'):
            return -pred.mean() if target_is_real else pred.mean()
        t = torch.full_like(pred, self.real if target_is_real else self.fake)
        if self.mode == 'lsgan':
            return l2_loss(pred, t, weight)
        if weight is None:   # reference's vanilla path multiplies by weight unconditionally (gan_loss.py:164)
            raise TypeError('vanilla GANLoss needs an explicit weight (reference: weight * loss with weight=None)')
        return (weight.float() * F.binary_cross_entropy_with_logits(pred, t, reduction='none')).sum()
