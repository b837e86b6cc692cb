"""Oracle of the training-batch preparation (TEST INFRASTRUCTURE ONLY; only tests / smoke / the bench's CPU leg may import it).

Restates, in numpy, what RDFC-GAN/lib/datasets/nyuv2/nyuv2_dataset_training.py:75-134 does to one decoded sample between
`cv2.imread` and the returned dict, i.e. the chain

    CutOffBlackBorder (preprocessing.py:312-327: rows 45:-15, columns 45:-40)
    TF.rotate(angle, resample=NEAREST)                      -> PIL Image.rotate -> Geometry.c affine_fixed (16.16 fixed point)
    T.Resize(height) (smaller edge -> height; BILINEAR,     -> PIL Image.resize -> Resample.c precompute_coeffs + two passes
              labels: NEAREST -> Geometry.c ImagingScaleAffine)   (8-bit images: 22-bit integer coefficients, clip8 after EACH pass;
    T.CenterCrop(crop)                                              'F' images: double accumulation, float between the passes)
    ToTensor (/255 for 8-bit), validity masks, T.Normalize

torchvision (0.7, README.md:63) is not installed in this image; the three transforms above are thin wrappers around the PIL calls
named on the right (torchvision/transforms/functional.py: rotate -> img.rotate(angle, resample, expand, center, fillcolor=None);
resize -> img.resize((ow, oh), interpolation) with ow / oh = int(size * w / h); center_crop -> crop at int(round((h - th) / 2.))).
Pinned: tests/test_batch_prep_cpu.py checks every stage of this restatement against PIL itself (Pillow is in the image), bit for bit.
"""
import math

import numpy as np

BORDER = (45, 15, 45, 40)          # top, bottom, left, right (preprocessing.py:313-317)
PRECISION_BITS = 32 - 8 - 2        # Resample.c


def cut_border(img):
    t, b, l, r = BORDER
    return img[t:-b, l:-r]


# ---- Image.rotate(angle, NEAREST): Image.py rotate() + Geometry.c affine_fixed ------------------------------------------------
def rotate_matrix(angle, w, h):
    """the six doubles Image.rotate hands to transform(AFFINE) (None: the 0 / 180-degree fast paths do not occur for random angles)"""
    angle = angle % 360.0
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def _fix(v):
    v = v * 65536.0 + 0.5
    return int(v) if v >= 0.0 else int(math.floor(v))


def fixed_matrix(m):
    """affine_fixed's integers: (a0, a1, a2, a3, a4, a5) with the half-pixel centre folded into a2 / a5"""
    return (_fix(m[0]), _fix(m[1]), _fix(m[2] + m[0] * 0.5 + m[1] * 0.5), _fix(m[3]), _fix(m[4]), _fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_source_index(fm, w, h):
    """(yin, xin, valid) per output pixel: xin = (a2 + y a1 + x a0) >> 16, yin = (a5 + y a4 + x a3) >> 16 (int32 arithmetic)"""
    a0, a1, a2, a3, a4, a5 = fm
    y, x = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing='ij')
    xin = (a2 + y * a1 + x * a0) >> 16
    yin = (a5 + y * a4 + x * a3) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    return yin, xin, ok


def rotate_nearest(img, angle):
    h, w = img.shape[:2]
    yin, xin, ok = rotate_source_index(fixed_matrix(rotate_matrix(angle, w, h)), w, h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


# ---- Image.resize: Resample.c ----------------------------------------------------------------------------------------------
def resize_size(w, h, size):
    """T.Resize(int): the smaller edge becomes `size`"""
    if (w <= h and w == size) or (h <= w and h == size):
        return w, h
    return (size, int(size * h / w)) if w < h else (int(size * w / h), size)


def bilinear_coeffs(in_size, out_size):
    """precompute_coeffs for the triangle filter over the whole axis: -> bounds [out, 2] (first tap, tap count), kk [out, ksize] (double)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w_ = 1.0 - t if t < 1.0 else 0.0
            kk[xx, x] = w_
            ww += w_
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def coeffs_8bpc(kk):
    """normalize_coeffs_8bpc: 22-bit fixed point, round half away from zero"""
    v = kk * (1 << PRECISION_BITS)
    return np.where(v < 0, (v - 0.5).astype(np.int64), (v + 0.5).astype(np.int64)).astype(np.int32)


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, ow, oh):
    """8-bit image [H, W, C]: horizontal pass, clip to uint8, vertical pass, clip to uint8"""
    h, w = img.shape[:2]
    src = img.astype(np.int64)
    if ow != w:
        bnd, kk = bilinear_coeffs(w, ow)
        k8 = coeffs_8bpc(kk)
        tmp = np.empty((h, ow) + img.shape[2:], dtype=np.uint8)
        for xx in range(ow):
            x0, n = bnd[xx]
            acc = np.full((h,) + img.shape[2:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for x in range(n):
                acc += src[:, x0 + x] * int(k8[xx, x])
            tmp[:, xx] = _clip8(acc)
        src = tmp.astype(np.int64)
        w = ow
    if oh != h:
        bnd, kk = bilinear_coeffs(h, oh)
        k8 = coeffs_8bpc(kk)
        out = np.empty((oh, w) + img.shape[2:], dtype=np.uint8)
        for yy in range(oh):
            y0, n = bnd[yy]
            acc = np.full((w,) + img.shape[2:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for y in range(n):
                acc += src[y0 + y] * int(k8[yy, y])
            out[yy] = _clip8(acc)
        return out
    return src.astype(np.uint8)


def resize_bilinear_f32(img, ow, oh):
    """'F' image [H, W]: double accumulation in tap order, float storage between and after the passes"""
    h, w = img.shape
    src = img.astype(np.float32)
    if ow != w:
        bnd, kk = bilinear_coeffs(w, ow)
        tmp = np.empty((h, ow), dtype=np.float32)
        for xx in range(ow):
            x0, n = bnd[xx]
            acc = np.zeros(h, dtype=np.float64)
            for x in range(n):
                acc += src[:, x0 + x].astype(np.float64) * kk[xx, x]
            tmp[:, xx] = acc.astype(np.float32)
        src, w = tmp, ow
    if oh != h:
        bnd, kk = bilinear_coeffs(h, oh)
        out = np.empty((oh, w), dtype=np.float32)
        for yy in range(oh):
            y0, n = bnd[yy]
            acc = np.zeros(w, dtype=np.float64)
            for y in range(n):
                acc += src[y0 + y].astype(np.float64) * kk[yy, y]
            out[yy] = acc.astype(np.float32)
        return out
    return src


def nearest_table(in_size, out_size):
    """ImagingScaleAffine's source index per output coordinate: xo = a/2, then xo += a per step (sequential double adds)"""
    a = in_size / out_size
    xo = a * 0.5
    tab = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        tab[x] = -1 if xo < 0.0 else int(xo)
        xo += a
    return tab


def resize_nearest(img, ow, oh):
    h, w = img.shape[:2]
    xt, yt = nearest_table(w, ow), nearest_table(h, oh)
    out = np.zeros((oh, ow) + img.shape[2:], dtype=img.dtype)
    okx, oky = (xt >= 0) & (xt < w), (yt >= 0) & (yt < h)
    out[np.ix_(oky, okx)] = img[np.ix_(yt[oky], xt[okx])]
    return out


def center_crop_origin(h, w, th, tw):
    return int(round((h - th) / 2.)), int(round((w - tw) / 2.))


def center_crop(img, th, tw):
    i, j = center_crop_origin(img.shape[0], img.shape[1], th, tw)
    return img[i:i + th, j:j + tw]


# ---- one sample, start to end (nyuv2_dataset_training.py:75-155) -----------------------------------------------------------------
def prepare_sample(rgb, gt_normal, raw_depth_mm, gt_depth_mm, labels, angle, height=240, crop=(224, 304),
                   rgb_mean=(0.485, 0.456, 0.406), rgb_std=(0.229, 0.224, 0.225), depth_mean=5.0, depth_std=5.0):
    """rgb, gt_normal: uint8 [480, 640, 3] (RGB order); raw / gt depth: uint16 millimetres [480, 640]; labels uint8 [480, 640];
    angle: the sample's draw of np.random.uniform(-5, 5).  -> the sample dict of get_train_data (numpy, CHW)"""
    f32 = np.float32
    raw = cut_border(raw_depth_mm.astype(f32) / f32(1000.0))           # (:73-74: float32 array / python float stays float32)
    gt = cut_border(gt_depth_mm.astype(f32) / f32(1000.0))
    rgb, gt_normal, labels = cut_border(rgb), cut_border(gt_normal), cut_border(labels)
    rot = lambda im: rotate_nearest(im, angle)
    rgb, gt_normal, raw, gt, labels = rot(rgb), rot(gt_normal), rot(raw), rot(gt), rot(labels)
    h, w = rgb.shape[:2]
    ow, oh = resize_size(w, h, height)
    to_t8 = lambda im: (center_crop(resize_bilinear_u8(im, ow, oh), *crop).transpose(2, 0, 1).astype(f32) / f32(255.0))
    to_tf = lambda im: center_crop(resize_bilinear_f32(im, ow, oh), *crop)[None]
    rgb_t, nrm_t = to_t8(rgb), to_t8(gt_normal)
    norm_mask = (nrm_t[0] > 0) | (nrm_t[1] > 0) | (nrm_t[2] > 0)
    lab = center_crop(resize_nearest(labels, ow, oh), *crop).astype(np.int64)
    raw_t, gt_t = to_tf(raw), to_tf(gt)
    mean, std = np.asarray(rgb_mean, f32).reshape(3, 1, 1), np.asarray(rgb_std, f32).reshape(3, 1, 1)
    depth_mask = gt_t > f32(0.0001)
    return dict(rgb=(rgb_t - mean) / std, gt_normal=(nrm_t - mean) / std, raw_depth=(raw_t - f32(depth_mean)) / f32(depth_std),
                gt_depth=(gt_t - f32(depth_mean)) / f32(depth_std), depth_masks=depth_mask, normal_masks=norm_mask, labels=lab)
