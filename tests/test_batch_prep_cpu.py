"""oracle/batch_prep.py (numpy restatement of the PIL / torchvision calls in nyuv2_dataset_training.py:75-134) pinned against
PIL itself, stage by stage and end to end, bit for bit.  torchvision is absent from the image; its rotate / Resize / CenterCrop /
ToTensor / Normalize are restated here with the PIL / torch calls they wrap (torchvision 0.7 transforms/functional.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import batch_prep as bp

Image = pytest.importorskip('PIL.Image')


def _sample(seed=0):
    r = np.random.default_rng(seed)
    sm = lambda a: np.array(Image.fromarray(a).resize((640, 480), Image.BILINEAR))          # smooth-ish content
    rgb = sm(r.integers(0, 256, (60, 80, 3), dtype=np.uint8))
    nrm = sm(r.integers(0, 256, (60, 80, 3), dtype=np.uint8))
    nrm[100:140, 200:300] = 0                                                                # an invalid-normal patch
    d = (r.random((480, 640)) * 9000 + 700).astype(np.uint16)
    d[300:330, 400:500] = 0                                                                   # invalid ground truth
    raw = d.copy()
    raw[200:260, 100:400] = 0
    lab = np.repeat(np.repeat(r.integers(0, 14, (40, 54), dtype=np.uint8), 12, 0), 12, 1)[:480, :640]
    return rgb, nrm, raw, d, lab


@pytest.mark.parametrize('angle', [3.7, -4.99, 0.31, -0.004])
def test_rotate_nearest_is_pils(angle):
    rgb, _, raw, _, lab = _sample()
    for arr, mode in ((bp.cut_border(rgb), 'RGB'), (bp.cut_border(raw.astype(np.float32) / np.float32(1000.0)), 'F'), (bp.cut_border(lab), 'L')):
        ref = np.asarray(Image.fromarray(np.ascontiguousarray(arr), mode=mode).rotate(angle, resample=Image.NEAREST))
        assert np.array_equal(bp.rotate_nearest(arr, angle), ref), mode


def test_resize_and_crop_are_pils():
    rgb, _, raw, _, lab = _sample(1)
    a8, af, al = bp.cut_border(rgb), bp.cut_border(raw.astype(np.float32) / np.float32(1000.0)), bp.cut_border(lab)
    h, w = a8.shape[:2]
    ow, oh = bp.resize_size(w, h, 240)
    assert (ow, oh) == (317, 240) and (h, w) == (420, 555)
    assert np.array_equal(bp.resize_bilinear_u8(a8, ow, oh), np.asarray(Image.fromarray(np.ascontiguousarray(a8), 'RGB').resize((ow, oh), Image.BILINEAR)))
    assert np.array_equal(bp.resize_bilinear_f32(af, ow, oh), np.asarray(Image.fromarray(np.ascontiguousarray(af), 'F').resize((ow, oh), Image.BILINEAR)))
    assert np.array_equal(bp.resize_nearest(al, ow, oh), np.asarray(Image.fromarray(np.ascontiguousarray(al), 'L').resize((ow, oh), Image.NEAREST)))
    assert bp.center_crop_origin(240, 317, 224, 304) == (8, 6)                  # int(round(6.5)) = 6: python's round-half-even, as upstream


def _reference_sample(rgb, nrm, raw_mm, gt_mm, lab, angle, height=240, crop=(224, 304)):
    """get_train_data (:75-155) with the torchvision calls spelled out as the PIL / torch calls they wrap"""
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    raw, gt = raw_mm.astype(np.float32) / 1000.0, gt_mm.astype(np.float32) / 1000.0
    cut = bp.cut_border
    ims = dict(rgb=Image.fromarray(np.ascontiguousarray(cut(rgb)), 'RGB'), nrm=Image.fromarray(np.ascontiguousarray(cut(nrm)), 'RGB'),
               raw=Image.fromarray(np.ascontiguousarray(cut(raw)), 'F'), gt=Image.fromarray(np.ascontiguousarray(cut(gt)), 'F'),
               lab=Image.fromarray(np.ascontiguousarray(cut(lab)), 'L'))
    ims = {k: v.rotate(angle, resample=Image.NEAREST) for k, v in ims.items()}

    def resize(im, interp):
        w, h = im.size
        ow, oh = (height, int(height * h / w)) if w < h else (int(height * w / h), height)
        return im.resize((ow, oh), interp)

    def ccrop(im):
        w, h = im.size
        i, j = int(round((h - crop[0]) / 2.)), int(round((w - crop[1]) / 2.))
        return im.crop((j, i, j + crop[1], i + crop[0]))

    def to_tensor(im):                                         # T.ToTensor on the numpy array upstream's ToNumpy yields
        a = np.asarray(im)
        t = torch.from_numpy(np.array(a if a.ndim == 3 else a[:, :, None]).transpose(2, 0, 1).copy())
        return t.float().div(255) if t.dtype == torch.uint8 else t

    def normalize(t, m, s):
        m, s = torch.tensor(m).view(-1, 1, 1), torch.tensor(s).view(-1, 1, 1)
        return t.clone().sub_(m).div_(s)

    rgb_t, nrm_t = (to_tensor(ccrop(resize(ims[k], Image.BILINEAR))) for k in ('rgb', 'nrm'))
    nmask = (nrm_t[0] > 0) | (nrm_t[1] > 0) | (nrm_t[2] > 0)
    labels = torch.from_numpy(np.array(ccrop(resize(ims['lab'], Image.NEAREST)))).long()
    raw_t, gt_t = (to_tensor(ccrop(resize(ims[k], Image.BILINEAR))) for k in ('raw', 'gt'))
    dmask = gt_t > 0.0001
    return dict(rgb=normalize(rgb_t, mean, std), gt_normal=normalize(nrm_t, mean, std), raw_depth=normalize(raw_t, [5.0], [5.0]),
                gt_depth=normalize(gt_t, [5.0], [5.0]), depth_masks=dmask, normal_masks=nmask, labels=labels)


@pytest.mark.parametrize('angle', [2.25, -3.9])
def test_prepare_sample_end_to_end(angle):
    s = _sample(2)
    ref = _reference_sample(*s, angle)
    got = bp.prepare_sample(*s, angle)
    assert set(got) == set(ref)
    for k, v in ref.items():
        assert got[k].shape == tuple(v.shape), k
        assert np.array_equal(got[k], v.numpy()), k
    assert got['labels'].dtype == np.int64 and got['depth_masks'].dtype == bool and got['rgb'].dtype == np.float32
    assert 0.02 < got['normal_masks'].mean() < 1.0 and not got['depth_masks'].all()         # the invalid patches survive the chain


def test_product_host_tables_equal_the_oracles():
    """rdfc_gan_amd.batch_prep builds the tap tables / fixed-point rotation integers the kernel consumes: same numbers as the
    (PIL-pinned) oracle's, for the shipped geometry and for up-scaling / identity axes"""
    from rdfc_gan_amd import batch_prep as P
    for n_in, n_out in ((555, 317), (420, 240), (640, 700), (100, 100)):
        b, k, _ = P._triangle_taps(n_in, n_out)
        ob, ok = bp.bilinear_coeffs(n_in, n_out)
        assert np.array_equal(b, ob) and np.array_equal(k, ok), (n_in, n_out)
        assert np.array_equal(P._fixed_coeffs(k), bp.coeffs_8bpc(ok))
        assert np.array_equal(P._nearest_taps(n_in, n_out), bp.nearest_table(n_in, n_out))
    for a in (3.7, -4.99, 0.0031, 359.2):
        assert P.rotation_fixed_point(a, 555, 420) == list(bp.fixed_matrix(bp.rotate_matrix(a, 555, 420)))
