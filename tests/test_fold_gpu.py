"""GPU: the in-launch second stage of the two-stage reductions (csrc/fold.h) -- the last-arriving workgroup finalizes.

Checked here: (1) the totals against an f64 host sum and against the separate finalize launches of the three-call API (same partial
rows, different but equally fixed association: equal to f32 rounding of an f64 sum), for one- and two-level folds, instance-norm
groups, stacked BatchNorm groups with running statistics, the scalar and the vector channel-sum paths; (2) bitwise reproducibility
run to run while another stream keeps the chip busy (uneven load: which workgroup arrives last changes, the result must not);
(3) that the arrival counters are left zeroed (many launches in a row stay correct)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def hip():
    from rdfc_gan_amd import hip as h
    return h


def _map(G, P, Cc, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(G * P, Cc, generator=g) * 1.5 + 0.3).to(dtype).to(DEV).view(1, 1, G * P, Cc)


# (G, P, C): nchunk <= 64 (one level), > 64 (two levels), the 4096-chunk cap, many instance-norm groups
SHAPES = [(1, 4 * 29 * 38, 512), (1, 16 * 57 * 76, 256), (1, 16 * 228 * 304, 64), (16, 57 * 76, 256), (2, 8 * 114 * 152, 128), (3, 1000, 8)]


@pytest.mark.parametrize('shape', SHAPES, ids=[f'g{g}p{p}c{c}' for g, p, c in SHAPES])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
def test_folded_statistics_match_f64_and_the_three_call_api(shape, dtype):
    h = hip()
    G, P, Cc = shape
    x = _map(G, P, Cc, dtype, 1)
    mean, rstd = h.norm_stats(x, G, 1e-5)
    xd = x.view(G, P, Cc).double()
    mu = xd.mean(1)
    var = xd.var(1, unbiased=False)
    assert torch.allclose(mean.double(), mu, rtol=0, atol=2e-6)
    assert torch.allclose(rstd.double(), (var + 1e-5).rsqrt(), rtol=3e-6, atol=0)
    # the same partial rows through the separate finalize launch
    L = h.lib()
    nch = h.norm_nchunk(P, Cc, h.dt(x))
    part = torch.empty(G * nch * 2 * Cc, dtype=torch.float32, device=DEV)
    m2, r2 = torch.empty(G, Cc, device=DEV), torch.empty(G, Cc, device=DEV)
    h._chk(L.rdfc_norm_stats(h._p(x), h._p(part), G, P, Cc, Cc, h.dt(x), h._st()), 'stats')
    h._chk(L.rdfc_norm_finalize(h._p(part), h._p(m2), h._p(r2), G, P, Cc, h.dt(x), 1e-5, None, None, 0.1, h._st()), 'finalize')
    assert torch.allclose(mean, m2, rtol=0, atol=1e-7) and torch.allclose(rstd, r2, rtol=2e-7, atol=0)


def test_folded_statistics_are_bitwise_reproducible_under_uneven_load():
    h = hip()
    G, P, Cc = 1, 16 * 114 * 152, 128
    x = _map(G, P, Cc, torch.float16, 2)
    busy = torch.randn(4096, 4096, device=DEV)
    side = torch.cuda.Stream()
    ref = None
    for it in range(24):
        if it % 2:                      # every other run shares the chip with a GEMM on another stream
            with torch.cuda.stream(side):
                for _ in range(3):
                    busy = (busy @ busy).clamp_(-1, 1)
        mean, rstd = h.norm_stats(x, G, 1e-5)
        got = torch.cat([mean.flatten(), rstd.flatten()]).clone()
        if ref is None:
            ref = got
        assert torch.equal(got, ref), f'run {it} differs'
    torch.cuda.synchronize()


@pytest.mark.parametrize('Cc', [64, 512])
def test_folded_stacked_batchnorm_running_statistics_in_group_order(Cc):
    """BatchNorm over 2 stacked batches (momenta 0.1 and 0.19): statistics per group, running statistics advanced group by group"""
    h = hip()
    G, P = 2, 8 * 57 * 76
    x = _map(G, P, Cc, torch.float16, 3)
    z = torch.empty_like(x)
    rm = torch.linspace(-1, 1, Cc, device=DEV).contiguous()
    rv = torch.linspace(0.5, 2, Cc, device=DEV).contiguous()
    rm0, rv0 = rm.clone().double(), rv.clone().double()
    moms = (0.1, 0.19)
    mean, rstd = h.norm_forward(x, G, 1e-5, None, None, None, z, 'none', 0.0, None, rm, rv, 0.1, group_momenta=moms)
    xd = x.view(G, P, Cc).double()
    for g in range(G):
        mu, var = xd[g].mean(0), xd[g].var(0, unbiased=False)
        assert torch.allclose(mean[g].double(), mu, atol=2e-6, rtol=0)
        rm0 = ((1 - moms[g]) * rm0 + moms[g] * mu).float().double()
        rv0 = ((1 - moms[g]) * rv0 + moms[g] * var * P / (P - 1)).float().double()
    assert torch.allclose(rm.double(), rm0, atol=1e-6, rtol=1e-6) and torch.allclose(rv.double(), rv0, atol=1e-6, rtol=1e-6)
    zr = ((xd - mean.double()[:, None]) * rstd.double()[:, None]).view_as(x)
    assert float((z.double() - zr).abs().max()) < 4e-3


@pytest.mark.parametrize('Cc', [1, 3, 64, 256])
def test_folded_channel_sum_and_accumulate(Cc):
    h = hip()
    P = 16 * 57 * 76 if Cc > 3 else 16 * 228 * 304
    x = _map(1, P, Cc, torch.float16, 4)
    acc = torch.full((Cc,), 2.0, device=DEV)
    for rep in range(3):                # the counters must come back zeroed: three launches in a row
        out = h.channel_sum(x.view(1, 1, P, Cc), acc=acc)
        ref = x.view(P, Cc).double().sum(0)
        assert torch.allclose(out.double(), ref, rtol=1e-6, atol=1e-3), (rep, float((out.double() - ref).abs().max()))
    assert torch.allclose(acc.double(), 2.0 + 3 * ref, rtol=1e-5, atol=1e-2)


def test_folded_backward_sums_against_autograd():
    """norm_backward (sums finalized in the reduce launch) against torch autograd of BatchNorm + LeakyReLU in f64"""
    h = hip()
    N, H, W, Cc = 4, 57, 76, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cc, generator=g).to(DEV)
    dz = torch.randn(N, H, W, Cc, generator=g).to(DEV)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    beta = torch.randn(Cc, generator=g).to(DEV)
    z = torch.empty_like(x)
    mean, rstd = h.norm_forward(x, 1, 1e-5, gamma, beta, None, z, 'lrelu', 0.2)
    dx, _, sums = h.norm_backward(dz, x, mean, rstd, gamma, beta, None, 'lrelu', 0.2)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mu = xr.mean((0, 1, 2))
    var = xr.var((0, 1, 2), unbiased=False)
    zr = torch.nn.functional.leaky_relu((xr - mu) * (var + 1e-5).rsqrt() * gr + br, 0.2)
    zr.backward(dz.double())
    assert float((dx.double() - xr.grad).abs().max() / xr.grad.abs().max()) < 2e-5
    assert torch.allclose(sums[0, 0].double(), br.grad, rtol=1e-5, atol=1e-4)
    assert torch.allclose(sums[0, 1].double(), gr.grad, rtol=1e-5, atol=1e-3)
