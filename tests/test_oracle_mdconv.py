"""CPU: pin the C restatement of modulated deformable conv (oracle/mdconv_oracle.c)
with the property tests of upstream's deformconv/test.py (which need CUDA there):
zero-offset == nn.Conv2d (:69-110), identity kernel (:142-181), finite-difference
gradcheck with offset ~ 2*N(0,1), input ~ 0.01*U(0,1), N=2,C=4,4x4, groups=2
(:405-434), plus an independent pure-torch autograd formulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mdconv as om


def _rand_case(seed, B=2, C=4, H=7, W=9, Co=6, k=3, s=1, p=1, d=1, g=2, dg=2, dtype=torch.float64, off_scale=2.0):
    gen = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = torch.randn(B, C, H, W, generator=gen, dtype=dtype)
    off = off_scale * torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=gen, dtype=dtype)
    msk = torch.sigmoid(torch.randn(B, dg * k * k, Ho, Wo, generator=gen, dtype=dtype)) * 2
    w = torch.randn(Co, C // g, k, k, generator=gen, dtype=dtype) * 0.3
    b = torch.randn(Co, generator=gen, dtype=dtype)
    return x, off, msk, w, b, dict(stride=s, padding=p, dilation=d, groups=g, dgroups=dg)


@pytest.mark.parametrize('s,p,d,g', [(1, 1, 1, 1), (2, 1, 1, 2), (1, 2, 2, 1), (1, 0, 1, 4)])
def test_zero_offset_is_conv2d(s, p, d, g):
    x, off, msk, w, b, cfg = _rand_case(0, s=s, p=p, d=d, g=g, dg=1, Co=8, C=4, dtype=torch.float32)
    off.zero_()
    msk.fill_(1.0)
    got = om.forward(x, off, msk, w, b, **cfg)
    ref = F.conv2d(x, w, b, s, p, d, g)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5)


def test_identity_kernel_reproduces_input():
    x, off, msk, _, _, cfg = _rand_case(1, C=4, Co=4, g=4, dg=1, dtype=torch.float32)
    off.zero_()
    msk.fill_(0.5)                                  # upstream: mask = sigmoid(0), output *= 2
    w = torch.zeros(4, 1, 3, 3)
    w[:, :, 1, 1] = 1.0
    got = om.forward(x, off, msk, w, torch.zeros(4), **cfg) * 2
    np.testing.assert_allclose(got.numpy(), x.numpy(), atol=1e-10)


@pytest.mark.parametrize('seed,kw', [(2, {}), (3, dict(s=2)), (4, dict(g=1, dg=1, C=1, Co=1)), (5, dict(k=1, p=0, g=1, dg=1, C=1, Co=1))])
def test_c_matches_torch_formulation_fwd_bwd(seed, kw):
    x, off, msk, w, b, cfg = _rand_case(seed, **kw)
    out_c = om.forward(x, off, msk, w, b, **cfg)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, msk, w, b)]
    out_t = om.forward_torch(*leaves, **cfg)
    np.testing.assert_allclose(out_c.numpy(), out_t.detach().numpy(), rtol=1e-10, atol=1e-10)
    go = torch.randn(out_t.shape, generator=torch.Generator().manual_seed(9), dtype=out_t.dtype)
    out_t.backward(go)
    grads_c = om.backward(x, off, msk, w, go, **cfg)
    for name, gc, leaf in zip(('input', 'offset', 'mask', 'weight', 'bias'), grads_c, leaves):
        np.testing.assert_allclose(gc.numpy(), leaf.grad.numpy(), rtol=1e-8, atol=1e-9, err_msg=name)


def test_finite_difference_gradcheck():
    """upstream check_gradient_mdconv (:405-434)."""
    gen = torch.Generator().manual_seed(11)
    N, C, H, W, Co, k, g, dg = 2, 4, 4, 4, 4, 3, 2, 2
    x = (0.01 * torch.rand(N, C, H, W, generator=gen, dtype=torch.float64)).requires_grad_(True)
    off = (2 * torch.randn(N, dg * 2 * k * k, H, W, generator=gen, dtype=torch.float64)).requires_grad_(True)
    msk = torch.sigmoid(torch.randn(N, dg * k * k, H, W, generator=gen, dtype=torch.float64)).requires_grad_(True)
    w = torch.randn(Co, C // g, k, k, generator=gen, dtype=torch.float64).requires_grad_(True)
    b = torch.rand(Co, generator=gen, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda *a: om.MDConvOracle.apply(*a, 1, 1, 1, g, dg, 2),
                                    (x, off, msk, w, b), eps=1e-6, atol=1e-5, rtol=1e-3, nondet_tol=0)


def test_out_of_range_samples_are_zero():
    x = torch.ones(1, 1, 4, 4)
    off = torch.zeros(1, 2, 4, 4)
    off[:, 0] = -5.0                                 # every sample lands above the image
    out = om.forward(x, off, torch.ones(1, 1, 4, 4), torch.ones(1, 1, 1, 1), torch.zeros(1), padding=0)
    assert float(out.abs().max()) == 0.0
    off[:, 0] = -0.5                                 # row -0.5: half weight from row 0 on the first row
    out = om.forward(x, off, torch.ones(1, 1, 4, 4), torch.ones(1, 1, 1, 1), torch.zeros(1), padding=0)
    np.testing.assert_allclose(out[0, 0, 0].numpy(), 0.5)
    np.testing.assert_allclose(out[0, 0, 1].numpy(), 1.0)
