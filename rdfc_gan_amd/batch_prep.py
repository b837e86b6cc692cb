"""Training-batch preparation on the device (`nyuv2_dataset_training.py:75-134`).

Upstream prepares every training sample on a dataloader worker with PIL: CutOffBlackBorder -> random rotation (nearest) ->
Resize (bilinear / nearest for labels) -> CenterCrop -> ToTensor -> masks -> Normalize.  `NYUv2BatchPrep` does the same for a
whole batch of decoded images in one launch of `rdfc_nyu_prepare_batch` and returns the dict `RDFCGAN.set_input` consumes
(same keys, shapes and dtypes as a collated batch of `get_train_data` samples) -- bit-identical to the host chain (the kernel
keeps PIL's integer / double arithmetic; see csrc/batch_prep.hip).  What stays on the host: decoding the files, and the
per-sample angle draw `np.random.uniform(-5.0, 5.0)` (:90) so that a seeded run consumes numpy's stream exactly as upstream.

The tap tables are functions of the geometry only (Resample.c precompute_coeffs, Geometry.c ImagingScaleAffine) and are
built once here; the six fixed-point integers of a rotation (Geometry.c affine_fixed) are computed per sample from its angle.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hip

_PBITS = 22


class PrepDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'Hraw', 'Wraw', 'top', 'left', 'Hc', 'Wc', 'Hr', 'Wr', 'cy0', 'cx0', 'Ho', 'Wo', 'ks')]


def _triangle_taps(n_in, n_out):
    """Resample.c precompute_coeffs (bilinear, whole axis) -> bounds int32 [n_out, 2], coefficients float64 [n_out, ks]"""
    scale = n_in / n_out
    fscale = max(scale, 1.0)
    support = fscale                                    # the triangle filter's support is 1.0
    ks = int(math.ceil(support)) * 2 + 1
    xx = np.arange(n_out, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # C truncation of non-negative doubles
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), n_in) - xmin
    j = np.arange(ks, dtype=np.float64)[None, :]
    t = np.abs((j + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    w = np.where(t < 1.0, 1.0 - t, 0.0)
    w[j >= xmax[:, None]] = 0.0
    # sequential sum in tap order, like the C loop (ww += w), then k /= ww
    ww = np.zeros(n_out)
    for q in range(ks):
        ww = ww + w[:, q]
    kk = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk, ks


def _fixed_coeffs(kk):
    """Resample.c normalize_coeffs_8bpc"""
    v = kk * float(1 << _PBITS)
    return np.where(v < 0, v - 0.5, v + 0.5).astype(np.int64).astype(np.int32)


def _nearest_taps(n_in, n_out):
    """Geometry.c ImagingScaleAffine: xo starts at a / 2 and grows by repeated addition"""
    a = n_in / n_out
    xo, tab = a * 0.5, np.empty(n_out, dtype=np.int32)
    for x in range(n_out):
        tab[x] = -1 if xo < 0.0 else int(xo)
        xo += a
    return tab


def rotation_fixed_point(angle, w, h):
    """Image.rotate's affine matrix for `angle` (degrees) about the centre of a w x h image, as affine_fixed's six integers"""
    angle = angle % 360.0
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(v) if v >= 0.0 else int(math.floor(v))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


class NYUv2BatchPrep:
    """prep = NYUv2BatchPrep(device); batch = prep(rgb, gt_normal, raw_depth_mm, gt_depth_mm, labels[, angles])"""

    def __init__(self, device, raw_hw=(480, 640), border=(45, 15, 45, 40), height=240, crop=(224, 304),
                 rgb_mean=(0.485, 0.456, 0.406), rgb_std=(0.229, 0.224, 0.225), depth_mean=5.0, depth_std=5.0, max_angle=5.0):
        self.device = torch.device(device)
        Hraw, Wraw = raw_hw
        top, bottom, left, right = border
        Hc, Wc = Hraw - top - bottom, Wraw - left - right
        # T.Resize(int): the smaller edge becomes `height`
        if (Wc <= Hc and Wc == height) or (Hc <= Wc and Hc == height):
            Wr, Hr = Wc, Hc
        elif Wc < Hc:
            Wr, Hr = height, int(height * Hc / Wc)
        else:
            Wr, Hr = int(height * Wc / Hc), height
        Ho, Wo = crop
        cy0, cx0 = int(round((Hr - Ho) / 2.)), int(round((Wr - Wo) / 2.))
        hb, hk, ksh = _triangle_taps(Wc, Wr)
        vb, vk, ksv = _triangle_taps(Hc, Hr)
        ks = max(ksh, ksv)
        pad = lambda k: np.pad(k, ((0, 0), (0, ks - k.shape[1])))
        hk, vk = pad(hk), pad(vk)
        dev = self.device
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        self._tabs = dict(hb=t(hb, torch.int32), hk8=t(_fixed_coeffs(hk), torch.int32), hkd=t(hk, torch.float64),
                          vb=t(vb, torch.int32), vk8=t(_fixed_coeffs(vk), torch.int32), vkd=t(vk, torch.float64),
                          xt=t(_nearest_taps(Wc, Wr), torch.int32), yt=t(_nearest_taps(Hc, Hr), torch.int32))
        self._ms = torch.tensor(list(rgb_mean) + list(rgb_std) + [depth_mean, depth_std], dtype=torch.float32, device=dev)
        self._geo = dict(Hraw=Hraw, Wraw=Wraw, top=top, left=left, Hc=Hc, Wc=Wc, Hr=Hr, Wr=Wr, cy0=cy0, cx0=cx0, Ho=Ho, Wo=Wo, ks=ks)
        self.max_angle = float(max_angle)

    def draw_angles(self, B):
        """the per-sample `degree = np.random.uniform(-5.0, 5.0)` of get_train_data (:90), in sample order"""
        return [float(np.random.uniform(-self.max_angle, self.max_angle)) for _ in range(B)]

    def __call__(self, rgb, gt_normal, raw_depth_mm, gt_depth_mm, labels, angles=None):
        """rgb / gt_normal: uint8 [B, Hraw, Wraw, 3] (RGB order, as after cv2.cvtColor); raw / gt depth: uint16 (int16 storage is
        accepted) millimetres [B, Hraw, Wraw]; labels uint8 [B, Hraw, Wraw]; host or device tensors / numpy arrays"""
        g, dev = self._geo, self.device

        def dv(a, dt):
            a = torch.as_tensor(a)
            if a.dtype == torch.uint16:
                a = a.view(torch.int16)
            assert a.dtype == dt, f'expected {dt}, got {a.dtype}'
            return a.to(dev, non_blocking=True).contiguous()
        rgb, gt_normal, labels = dv(rgb, torch.uint8), dv(gt_normal, torch.uint8), dv(labels, torch.uint8)
        raw, gt = dv(raw_depth_mm, torch.int16), dv(gt_depth_mm, torch.int16)
        B = rgb.shape[0]
        assert tuple(rgb.shape) == (B, g['Hraw'], g['Wraw'], 3) == tuple(gt_normal.shape)
        assert tuple(raw.shape) == (B, g['Hraw'], g['Wraw']) == tuple(gt.shape) == tuple(labels.shape)
        if angles is None:
            angles = self.draw_angles(B)
        aff = torch.tensor([rotation_fixed_point(a, g['Wc'], g['Hc']) for a in angles], dtype=torch.int32).to(dev, non_blocking=True)
        Ho, Wo = g['Ho'], g['Wo']
        out = dict(rgb=torch.empty(B, 3, Ho, Wo, dtype=torch.float32, device=dev), gt_normal=torch.empty(B, 3, Ho, Wo, dtype=torch.float32, device=dev),
                   raw_depth=torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=dev), gt_depth=torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=dev),
                   depth_masks=torch.empty(B, 1, Ho, Wo, dtype=torch.bool, device=dev), normal_masks=torch.empty(B, Ho, Wo, dtype=torch.bool, device=dev),
                   labels=torch.empty(B, Ho, Wo, dtype=torch.int64, device=dev))
        d = PrepDesc(B, g['Hraw'], g['Wraw'], g['top'], g['left'], g['Hc'], g['Wc'], g['Hr'], g['Wr'], g['cy0'], g['cx0'], Ho, Wo, g['ks'])
        T, p = self._tabs, hip._p
        hip._chk(hip.lib().rdfc_nyu_prepare_batch(C.byref(d), p(rgb), p(gt_normal), p(raw), p(gt), p(labels), p(aff), p(T['hb']), p(T['hk8']),
                                                  p(T['hkd']), p(T['vb']), p(T['vk8']), p(T['vkd']), p(T['xt']), p(T['yt']), p(self._ms),
                                                  p(out['rgb']), p(out['gt_normal']), p(out['raw_depth']), p(out['gt_depth']),
                                                  p(out['depth_masks']), p(out['normal_masks']), p(out['labels']), hip._st()),
                 'rdfc_nyu_prepare_batch')
        return out
