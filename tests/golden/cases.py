"""Shared, reference-free definitions of the golden test cases: deterministic
inputs, the linear output functionals used for backward, summaries.  Imported by
make_golden.py (build container) and by the tests (everywhere)."""
import numpy as np
import torch
import torch.nn.functional as F

from detweights import det_tensor

NLSPN_CFG = dict(prop_kernel=3, prop_time=18, affinity='TGASS', affinity_gamma=0.5,
                 conf_prop=True, preserve_input=False)

STEP_HP = dict(B=2, H=60, W=76, lr=2e-3, beta1=0.5, beta2=0.999, lambda_A=100.0, lambda_B=100.0, lambda_L1=100.0)


def _smooth(t, k=9):
    c = t.shape[1]
    w = torch.ones(c, 1, k, k) / (k * k)
    return F.conv2d(F.pad(t, (k // 2,) * 4, mode='replicate'), w, groups=c)


def synth_depth(seed, key, B, H, W):
    """NYUv2-like metric depth (mean 2.84 m, std 0.99 m, clipped), normalised (m-5)/5;
    raw depth = gt with rectangular holes set to 0 m -> -1.0."""
    m = (2.84 + 3.0 * _smooth(det_tensor(seed, key + '.m', (B, 1, H, W)))).clamp(0.713, 9.995)
    gt = (m - 5.0) / 5.0
    raw = gt.clone()
    r = det_tensor(seed, key + '.holes', (B, 6, 4), 'rand')
    for b in range(B):
        for y0, x0, hh, ww in r[b].tolist():
            ys, xs = int(y0 * H * 0.8), int(x0 * W * 0.8)
            raw[b, :, ys:ys + max(2, int(hh * H * 0.35)), xs:xs + max(2, int(ww * W * 0.35))] = -1.0
    mask = m > 1e-4
    return gt, raw, mask


def unit_normals(seed, key, B, H, W):
    return F.normalize(_smooth(det_tensor(seed, key, (B, 3, H, W)), 5) + 0.05 * det_tensor(seed, key + '.n', (B, 3, H, W)), dim=1)


def gen_inputs(tag, B, H, W):
    rgb = det_tensor(1, tag + '.rgb', (B, 3, H, W), 'rand') * 2 - 1
    _, raw, _ = synth_depth(1, tag + '.depth', B, H, W)
    return rgb, raw, unit_normals(1, tag + '.normal', B, H, W)


def out_weights(tag, key, shape):
    """Fixed random cotangent so that backward exercises every output."""
    return det_tensor(7, f'{tag}.{key}', tuple(shape))


def grad_summary(net):
    names, rows = [], []
    for k, p in net.named_parameters():
        if p.grad is not None:
            names.append(k)
            g = p.grad.double()
            rows.append([float(g.sum()), float(g.norm())])
    return names, np.array(rows)


def step_batch(it, B, H, W):
    gt, raw, mask = synth_depth(20 + it, 'step.depth', B, H, W)
    data = dict(rgb=det_tensor(20 + it, 'step.rgb', (B, 3, H, W), 'rand') * 2 - 1,
                gt_depth=gt, raw_depth=raw, depth_masks=mask,
                gt_normal=unit_normals(20 + it, 'step.gtn', B, H, W),
                labels=torch.zeros(B, H, W, dtype=torch.long),
                normal_masks=torch.ones(B, 1, H, W))
    return data, unit_normals(20 + it, 'step.nA', B, H, W), unit_normals(20 + it, 'step.nfA', B, H, W)


def pepper_mask(seed, shape):
    """The mask upstream's addPepperNoise would draw after random.seed(seed) /
    np.random.seed(seed) (helper.py:384-402): returns None (no noise this call) or
    an int array in {0,1,2} of shape (B,1,H,W)."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    if not random.uniform(0, 1) < 0.9:
        return None
    b, c, h, w = shape
    snr = 0.98
    pct = 1 - snr
    return np.random.choice((0, 1, 2), size=(b, 1, h, w), p=[snr, pct / 2., pct / 2.])


def noise_driven(net_key, param_key):
    """Parameters whose gradient is analytically zero (a conv bias in front of an InstanceNorm in
    G_B2A): autograd returns rounding noise for them and Adam turns that noise into lr-sized
    updates, upstream included -- their trajectories are not comparable between implementations
    (and they cannot influence any output)."""
    return net_key == 'G_B2A' and param_key.endswith('.bias') and param_key != 'model.23.bias'


def metric_case(B=3, H=40, W=52, seed=77):
    """normalised prediction / ground truth in [-1, 1] as the evaluator sees them (evaluator.py:17-46 de-normalises with
    std = mean = 5), with invalid ground-truth pixels (0 m -> -1 after normalisation) and an evaluate_mask"""
    g = torch.Generator().manual_seed(seed)
    gt_m = (torch.rand(B, 1, H, W, generator=g) * 9.0 + 0.7)
    hole = torch.rand(B, 1, H, W, generator=g) < 0.2
    gt_m[hole] = 0.0
    pred_m = (gt_m + torch.randn(B, 1, H, W, generator=g) * 0.4).clamp_min(0.0)
    pred_m[torch.rand(B, 1, H, W, generator=g) < 0.05] = 0.0        # predictions at / below t_valid exercise the inverse-depth rule
    ev = torch.rand(B, 1, H, W, generator=g) < 0.9
    return (gt_m - 5.0) / 5.0, (pred_m - 5.0) / 5.0, ev


def label_net_weights(template):
    """deterministic weights for G_label (PSPNet / dilated ResNet-50, 53 train-mode BatchNorm layers deep): the per-key
    draws of detweights with the last BatchNorm of every residual branch scaled by 0.1.  With gamma ~ 1 there the
    train-mode gradient grows by ~1.4x per block and upstream's own f32 input gradient is 1.6e-2 (relative L2) away from
    an f64 evaluation -- nothing could be pinned against it; with 0.1 it is 6e-4."""
    from detweights import det_state_dict
    sd = det_state_dict(template, 0)
    for k in sd:
        if k.endswith('bn3.weight'):
            sd[k] = sd[k] * 0.1
    return sd


LABEL_CASE = dict(B=4, H=48, W=64, classes=14)


def normal_loss_case(B=3, H=30, W=38, seed=31, classes=14):
    """predicted / ground-truth unit normals, a normal mask and class logits for nor_loss / manhattan_loss.  A patch of
    predictions equals the ground truth (cos = 1: excluded by the 0.999 rule) and a patch faces exactly (0, 1, 0) (a floor
    pixel whose 1 - cos is exactly 0 is not counted by upstream's `!= 0`)."""
    pred = unit_normals(seed, 'nl.pred', B, H, W)
    gt = unit_normals(seed, 'nl.gt', B, H, W)
    pred[:, :, :4, :6] = gt[:, :, :4, :6]
    pred[:, :, 10:14, 10:20] = torch.tensor([0.0, 1.0, 0.0]).view(1, 3, 1, 1)
    mask = (det_tensor(seed, 'nl.mask', (B, 1, H, W), 'rand') > 0.2).float()
    logits = _smooth(det_tensor(seed, 'nl.logits', (B, classes, H, W)), 3) * 4
    logits[:, 5, 9:16, 8:24] += 3.0                                   # a floor region over the exact-up patch
    return pred, gt, mask, logits


def step_batch_label(it, B, H, W, classes=14):
    """step_batch with real label maps (blocky, every class present) and a normal mask with holes"""
    data, nA, nfA = step_batch(it, B, H, W)
    z = _smooth(det_tensor(40 + it, 'step.labels', (B, classes, H, W)), 7)
    data['labels'] = z.argmax(1)
    data['normal_masks'] = (det_tensor(40 + it, 'step.nmask', (B, 1, H, W), 'rand') > 0.15).float()
    return data, nA, nfA


NORMAL_FEATS = {4: (24, 32, 40), 5: (40, 16, 20), 6: (64, 8, 10), 8: (176, 4, 5), 11: (2048, 2, 3)}   # index -> (C, h, w)


def normal_features(B=2, seed=51):
    """the five EfficientNet feature maps G_normal's decoder consumes (decoder.py:17), as seeded tensors"""
    return [det_tensor(seed, f'nfeat{i}', (B, c, h, w)) for i, (c, h, w) in NORMAL_FEATS.items()]


class FeatureListAdapter(torch.nn.Module):
    """five tensors -> upstream's 12-entry feature list -> decoder; parameters under 'n.' like the other fixtures"""

    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, f4, f5, f6, f8, f11):
        feats = [None] * 12
        for i, t in zip(NORMAL_FEATS, (f4, f5, f6, f8, f11)):
            feats[i] = t
        return self.n(feats)


class StandInEncoder(torch.nn.Module):
    """Stock-PyTorch stand-in for upstream's torch.hub EfficientNet-B5 encoder (normal_net/submodules/encoder.py:7-31, not in the
    tree): the same contract -- a 12-entry feature list whose entries 4, 5, 6, 8, 11 carry 24 / 40 / 64 / 176 / 2048 channels at
    1/4 ... 1/32 of the input -- from five small convolutions.  Used on BOTH sides of the G_normal step trace."""

    def __init__(self):
        super().__init__()
        self.convs = torch.nn.ModuleList(torch.nn.Conv2d(3, c, 3, padding=1) for c, _, _ in NORMAL_FEATS.values())
        self.strides = (4, 8, 16, 32, 32)

    def forward(self, img):
        feats = [None] * 12
        for i, conv, s in zip(NORMAL_FEATS, self.convs, self.strides):
            feats[i] = torch.tanh(conv(F.avg_pool2d(img, s, s, ceil_mode=True)))
        return feats
