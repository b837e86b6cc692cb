"""Operators of the semantic-label net (rdfc_gan_amd/csrc/label_ops.hip) through the C ABI against torch's CPU operators
(the ones upstream's PSPNet calls: labels_model.py:12-94, backbone_resnet50.py:172; loss rdfc_gan.py:59)."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import DEV, DTYPES, from_pm, hip, rel_err, rnd, to_pm, tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', DTYPES[:2], ids=['f32', 'bf16'])     # (the label net is not part of the f16 config)
@pytest.mark.parametrize('shape', [(2, 16, 17, 21), (1, 128, 114, 152), (2, 8, 5, 4)], ids=str)
def test_maxpool_forward_backward(shape, dtype):
    from rdfc_gan_amd import functional as Fh
    N, C, H, W = shape
    x = rnd(N, C, H, W, seed=1).to(dtype).float()
    x = torch.relu(x)                                        # ties at 0 like behind the stem's ReLU: first index must win
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    gy = rnd(*y_ref.shape, seed=2).to(dtype).float()
    y_ref.backward(gy)
    xp = to_pm(x, dtype, ld_extra=8).requires_grad_(True)
    y = Fh.max_pool2d(xp, 3, 2, 1)
    assert torch.equal(from_pm(y.detach()), y_ref.detach())
    gx, = torch.autograd.grad(y, xp, to_pm(gy, dtype))
    assert rel_err(from_pm(gx), xr.grad) < (1e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize('dtype', DTYPES[:2], ids=['f32', 'bf16'])     # (the label net is not part of the f16 config)
@pytest.mark.parametrize('bins', [1, 2, 3, 6])
def test_adaptive_avgpool(bins, dtype):
    from rdfc_gan_amd import functional as Fh
    N, C, H, W = 2, 64, 29, 38
    x = rnd(N, C, H, W, seed=3).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    y_ref = F.adaptive_avg_pool2d(xr, bins)
    gy = rnd(*y_ref.shape, seed=4).to(dtype).float()
    y_ref.backward(gy)
    xp = to_pm(x, dtype).requires_grad_(True)
    y = Fh.adaptive_avg_pool2d(xp, bins)
    assert rel_err(from_pm(y.detach()), y_ref.detach()) < (2e-6 if dtype == torch.float32 else 8e-3)
    gx, = torch.autograd.grad(y, xp, to_pm(gy, dtype))
    assert rel_err(from_pm(gx), xr.grad) < (2e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize('dtype', DTYPES[:2], ids=['f32', 'bf16'])     # (the label net is not part of the f16 config)
@pytest.mark.parametrize('case', [(2, 14, 29, 38, 228, 304), (2, 16, 1, 1, 29, 38), (1, 8, 2, 2, 29, 38), (2, 8, 3, 3, 9, 11),
                                  (1, 8, 6, 6, 29, 38), (2, 5, 7, 9, 7, 9), (1, 3, 12, 10, 5, 4)], ids=str)
def test_bilinear_align_corners(case, dtype):
    from rdfc_gan_amd import functional as Fh
    N, C, H, W, Ho, Wo = case
    x = rnd(N, C, H, W, seed=5).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, size=(Ho, Wo), mode='bilinear', align_corners=True)
    gy = rnd(*y_ref.shape, seed=6).to(dtype).float()
    y_ref.backward(gy)
    xp = to_pm(x, dtype).requires_grad_(True)
    for out_f32 in (False, True):
        y = Fh.interpolate_bilinear(xp, (Ho, Wo), out_f32=out_f32)
        assert y.dtype == (torch.float32 if out_f32 else dtype)
        assert rel_err(from_pm(y.detach()), y_ref.detach()) < (1e-5 if (dtype == torch.float32 or out_f32) else 8e-3)   # f32: fma contraction vs the CPU kernel
        gyp = to_pm(gy, torch.float32 if out_f32 else dtype)
        gx, = torch.autograd.grad(y, xp, gyp)
        assert gx.dtype == dtype
        assert rel_err(from_pm(gx), xr.grad) < (1e-5 if dtype == torch.float32 else 8e-3)


def test_bilinear_into_channel_window():
    """the pyramid levels are written straight into their slice of the concatenated map (labels_model.py:36-39)"""
    h = hip()
    x = rnd(2, 16, 3, 3, seed=7)
    wide = torch.zeros(2, 29, 38, 48, device=DEV)
    h.bilinear(to_pm(x, torch.float32), 29, 38, out=wide[..., 16:32])
    ref = F.interpolate(x, size=(29, 38), mode='bilinear', align_corners=True)
    assert rel_err(from_pm(wide[..., 16:32]), ref) < 2e-6
    assert float(wide[..., :16].abs().max()) == 0.0 and float(wide[..., 32:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', DTYPES[:2], ids=['f32', 'bf16'])     # (the label net is not part of the f16 config)
def test_dropout2d_scale(dtype):
    from rdfc_gan_amd import functional as Fh
    x = rnd(3, 32, 7, 9, seed=8).to(dtype).float()
    xp = to_pm(x, dtype).requires_grad_(True)
    torch.manual_seed(11)
    y = Fh.dropout2d(xp, 0.25, True)
    yv = from_pm(y.detach())
    ratio = (yv / x.clamp_min(1e-30).where(x != 0, torch.ones(()))).nan_to_num()
    per = yv.abs().amax(dim=(2, 3))
    dropped = per == 0
    assert 0 < int(dropped.sum()) < dropped.numel()                      # some (n, c) planes gone, not all
    keep = (~dropped)[:, :, None, None].float() / 0.75
    assert rel_err(yv, x * keep) < (1e-6 if dtype == torch.float32 else 8e-3)
    gx, = torch.autograd.grad(y, xp, torch.ones_like(y))
    assert rel_err(from_pm(gx), keep.expand_as(x)) < (1e-6 if dtype == torch.float32 else 8e-3)
    assert Fh.dropout2d(xp, 0.25, False) is xp


@pytest.mark.parametrize('dtype', DTYPES[:2], ids=['f32', 'bf16'])     # (the label net is not part of the f16 config)
@pytest.mark.parametrize('case', [(2, 14, 19, 23, 0.0), (2, 14, 19, 23, 0.3), (1, 14, 228, 304, 0.1), (1, 3, 5, 5, 0.5)], ids=str)
def test_cross_entropy(case, dtype):
    from rdfc_gan_amd import functional as Fh
    N, C, H, W, ignored = case
    z = (rnd(N, C, H, W, seed=9) * 3).to(dtype).float()
    g = torch.Generator().manual_seed(10)
    t = torch.randint(0, C, (N, H, W), generator=g)
    t[torch.rand(N, H, W, generator=g) < ignored] = -100
    zr = z.clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss()(zr, t)
    (ref * 1.7).backward()
    zp = to_pm(z, dtype).requires_grad_(True)
    loss = Fh.cross_entropy(zp, t.to(DEV))
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6 * max(1.0, abs(float(ref.detach())))
    gz, = torch.autograd.grad(loss * 1.7, zp)
    assert rel_err(from_pm(gz), zr.grad) < (3e-6 if dtype == torch.float32 else 8e-3)


def test_bad_arguments_are_reported():
    h = hip()
    x = torch.zeros(1, 4, 4, 6, device=DEV)                 # 6 channels: not a multiple of the 16-byte vector
    with pytest.raises(RuntimeError, match='multiple of'):
        h.maxpool2d(x, 3, 2, 1)
    with pytest.raises(RuntimeError, match='bad shape'):
        h.cross_entropy(torch.zeros(1, 2, 2, 5000, device=DEV), torch.zeros(1, 2, 2, dtype=torch.int64, device=DEV))


def test_normal_losses_against_upstream_fixture_and_oracle(golden):
    """rdfc_normal_losses: nor_loss + manhattan_loss values (gan_loss.py:122-132, 50-99) against the fixture from upstream's
    functions and, on a full-size batch, against the oracle"""
    from oracle import losses as ol
    from cases import normal_loss_case
    import numpy as np
    h = hip()
    g = golden('normal_losses')
    pred, gt, mask, logits = normal_loss_case()
    z = logits.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2)          # the label net's channels-last view
    out = h.normal_losses(pred.to(DEV), gt.to(DEV), mask.to(DEV), z, 12, 5, 3).cpu().numpy()
    np.testing.assert_allclose(out * [1, 100, 100, 100], [g['nor'], g['floor'], g['wall'], g['ceiling']], rtol=3e-6)
    out = h.normal_losses(pred.to(DEV), gt.to(DEV), None, z, 12, 5, 99).cpu().numpy()
    np.testing.assert_allclose(out[0], g['nor_nomask'], rtol=3e-6)
    assert out[3] == 0.0                                                              # no pixel of that class
    pred, gt, mask, logits = normal_loss_case(B=8, H=228, W=304, seed=32)
    z = logits.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2)
    out = h.normal_losses(pred.to(DEV), gt.to(DEV), mask.to(DEV), z, 12, 5, 3).cpu().numpy()
    f, w, c = ol.manhattan_loss(pred, logits, 12, 5, 3, 1.0)
    np.testing.assert_allclose(out, [float(ol.nor_loss(pred, gt, mask)), float(f), float(w), float(c)], rtol=2e-5)


def test_normal_losses_gradient(golden):
    """d(200 nor + 100 (floor + wall + ceiling)) / d normal: fixture from upstream's autograd, and the oracle's autograd on a
    full-size batch with un-normalised predictions (|a| != 1 exercises the 1 / |a| factors)"""
    from oracle import losses as ol
    from cases import normal_loss_case
    from rdfc_gan_amd import functional as Fh
    import numpy as np
    g = golden('normal_losses')
    wts = torch.tensor([200.0, 100.0, 100.0, 100.0], device=DEV)

    def hip_grad(pred, gt, mask, logits):
        z = logits.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2)
        pr = pred.to(DEV).requires_grad_(True)
        out = Fh.normal_losses(pr, gt.to(DEV), mask.to(DEV), z, 12, 5, 3)
        (out * wts).sum().backward()
        return pr.grad.cpu()

    pred, gt, mask, logits = normal_loss_case()
    got = hip_grad(pred, gt, mask, logits)
    ref = torch.from_numpy(g['grad_pred'])
    assert float((got - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    pred, gt, mask, logits = normal_loss_case(B=4, H=114, W=152, seed=33)
    pred = pred * (0.5 + torch.rand(4, 1, 114, 152, generator=torch.Generator().manual_seed(3)))
    got = hip_grad(pred, gt, mask, logits)
    pr = pred.clone().requires_grad_(True)
    f, w, c = ol.manhattan_loss(pr, logits, 12, 5, 3, 100.0)
    (ol.nor_loss(pr, gt, mask) * 200.0 + f + w + c).backward()
    assert float((got - pr.grad).abs().max()) < 2e-5 * float(pr.grad.abs().max())
