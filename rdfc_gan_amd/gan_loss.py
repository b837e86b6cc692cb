"""Loss stack with upstream's names and call signatures, on the fused HIP reduction.

L1_loss :6-22, L2_loss :25-42, mse_loss :104-120, GANLoss :169-206 of
RDFC-GAN/lib/losses/gan_loss.py.  Each call is one kernel producing the weighted sum
and, when the prediction needs a gradient, d(loss)/d(pred) in the same pass.
(nor_loss / manhattan_loss / vanilla BCE belong to the G_normal / G_label terms, which
are outside the hot path.)
"""
import torch
import torch.nn as nn

from . import functional as Fh


def _prep_weight(pred, weight):
    if weight is None:
        return None
    weight = weight.float()
    if weight.dim() != pred.dim():
        weight = weight.unsqueeze(1)
    if weight.shape[1] != 1 and weight.shape[1] == pred.shape[1] and pred.shape[1] != 1:
        raise NotImplementedError('per-channel weights are not used on the hot path (weight is (B,1,H,W))')
    return weight


def L1_loss(pred, target, weight=None, reduction='sum'):
    assert reduction == 'sum'
    return Fh.weighted_loss(pred, target, _prep_weight(pred, weight), 1)


def L2_loss(pred, target, weight=None, reduction='sum'):
    assert reduction == 'sum'
    return Fh.weighted_loss(pred, target, _prep_weight(pred, weight), 2)


mse_loss = L2_loss


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer('real_label', torch.tensor([target_real_label]))
        self.register_buffer('fake_label', torch.tensor([target_fake_label]))
        self.gan_mode = gan_mode
        if gan_mode not in ('lsgan', 'wgan', 'wgangp'):
            raise NotImplementedError(f'gan mode {gan_mode} not implemented on the HIP path (lsgan, wgan, wgangp)')
        self._labels = (float(target_real_label), float(target_fake_label))

    def __call__(self, prediction, target_is_real, weight=None):
        if self.gan_mode == 'lsgan':
            # the constant target is folded into the kernel instead of an expand_as tensor (gan_loss.py:187-192)
            return Fh.weighted_loss(prediction, self._labels[0] if target_is_real else self._labels[1],
                                    _prep_weight(prediction, weight), 2)
        return -prediction.mean() if target_is_real else prediction.mean()
