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


def _cos(a, b):
    """torch.cosine_similarity(a, b, dim=1) as ATen evaluates it: (a / max(|a|, eps)) . (b / max(|b|, eps)), eps = 1e-8"""
    an = a.norm(dim=1, keepdim=True).clamp_min(1e-8)
    bn = b.norm(dim=1, keepdim=True).clamp_min(1e-8)
    return ((a / an) * (b / bn)).sum(1)


def nor_loss(pred, target, norm_masks):
    """lib/losses/gan_loss.py:122-132: mean of 1 - cos over the masked pixels with |cos| < 0.999"""
    dot = _cos(pred, target)
    if norm_masks.dim() != 4:
        norm_masks = norm_masks.unsqueeze(1)
    valid = norm_masks[:, 0].float() * (dot.detach() < 0.999).float() * (dot.detach() > -0.999).float() > 0
    val = (1.0 - dot) * valid
    return val.sum() / ((val != 0).sum() + 1e-6)


def manhattan_loss(normal, semantic, label_wall, label_floor, label_ceiling, weight):
    """lib/losses/gan_loss.py:50-99 -> (floor, wall, ceiling) * weight: the arg-max class picks the pixels; floors should
    face (0, 1, 0), ceilings (0, -1, 0), walls be perpendicular to it.

    Upstream multiplies the per-image cosine map [B,H,W] by the unsqueezed mask [B,1,H,W] (:73-74, 83, 93): the product
    broadcasts to [B,B,H,W] -- image j's term at a pixel enters once for every image i whose pixel has the class.  Written
    out here with explicit sums (no broadcasting) so that the quirk is visible: M = per-pixel count of images with the
    class, terms summed over j, both weighted by M."""
    cls = semantic.argmax(1)                                        # softmax is monotone: same arg-max (:52)
    up = torch.zeros_like(normal)
    up[:, 1] = 1.0
    out = []
    for label, kind in ((label_floor, 'floor'), (label_wall, 'wall'), (label_ceiling, 'ceiling')):
        m = (cls == label)                                          # [B,H,W]
        if int(m.sum()) == 0:
            out.append(torch.zeros(()))
            continue
        c = _cos(normal * m.unsqueeze(1), -up if kind == 'ceiling' else up)      # 0 outside image j's mask
        t = (c * 2).abs() if kind == 'wall' else (c * -1 + 1)       # [B,H,W]; floor / ceiling: 1 outside the mask
        M = m.sum(0).to(t.dtype)                                    # [H,W]
        total = (M * t.sum(0)).sum()
        count = (M * (t != 0).sum(0)).sum()
        out.append(total / (count + 1e-6) * weight)
    return tuple(out)
