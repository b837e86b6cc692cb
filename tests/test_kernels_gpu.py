"""GPU: every HIP kernel behind the C ABI against a plain PyTorch fp32 CPU reference
of the same op (kernel-level parity; module-level parity against the oracle and the
golden fixtures is in test_modules_gpu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
DT_IDS = ['f32', 'bf16', 'f16']


def hip():
    from rdfc_gan_amd import hip as h
    return h


def tol(dtype):
    """relative to the largest reference value; operands are rounded identically on both sides, so what is left is the
    rounding of the result: 2^-9 (bf16: 8 significant bits) / 2^-12 (f16: 11 bits) per element, plus summation order"""
    return {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dtype]


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_pm(x_nchw, dtype, ld_extra=0):
    """NCHW cpu f32 -> pixel-major device tensor (optionally a channel slice of a wider map)."""
    N, C, H, W = x_nchw.shape
    v = x_nchw.permute(0, 2, 3, 1).contiguous().to(dtype)
    if ld_extra:
        buf = torch.full((N, H, W, C + ld_extra), 7.0, dtype=dtype)
        buf[..., :C] = v
        return buf.to(DEV)[..., :C]
    return v.to(DEV)


def from_pm(x_pm):
    return x_pm.float().cpu().permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, W, Ci, Co, k, s, p
    (2, 29, 38, 512, 14, 1, 1, 0),          # label net: class logits (labels_model.py:63,72)
    (2, 20, 24, 3, 64, 3, 2, 1),            # label net: deep stem (backbone_resnet50.py:155)
    (1, 9, 11, 4096, 512, 1, 1, 0),         # label net: pyramid bottleneck (labels_model.py:19)
    (2, 12, 14, 64, 64, 3, 1, 1),
    (2, 13, 15, 64, 128, 3, 2, 1),
    (1, 9, 11, 128, 256, 1, 2, 0),
    (2, 10, 12, 128, 1, 3, 1, 1),
    (2, 10, 12, 96, 1, 3, 1, 1),
    (2, 10, 12, 128, 32, 3, 1, 1),
    (2, 10, 12, 3, 48, 3, 1, 1),
    (2, 10, 12, 1, 16, 3, 1, 1),
    (2, 16, 18, 3, 64, 4, 2, 1),
    (2, 16, 18, 1, 64, 4, 2, 1),
    (2, 9, 9, 512, 1, 4, 1, 1),
    (1, 20, 22, 1, 64, 7, 1, 0),
    (1, 20, 22, 64, 3, 7, 1, 0),
    (1, 7, 8, 768, 128, 3, 1, 1),
    (2, 15, 19, 256, 512, 4, 1, 1),
    (3, 29, 38, 64, 64, 3, 1, 1),
    (2, 8, 8, 8, 24, 3, 1, 1),
    # tap-list gather kernel (conv_gather_glds.hip): 4x4 / s2 input-gradient classes, 1x1, odd sizes
    (2, 16, 18, 64, 128, 4, 2, 1),
    (2, 17, 21, 128, 64, 3, 2, 1),
    (2, 10, 12, 192, 384, 1, 1, 0),
    (1, 19, 23, 64, 64, 4, 1, 1),
    # gather-based LDS-DMA weight gradient (conv_wgrad_gather_glds.hip) needs rows of >= ~18 output pixels
    (2, 30, 76, 64, 128, 3, 2, 1),
    (2, 32, 128, 64, 128, 4, 2, 1),
    (1, 20, 60, 128, 256, 4, 1, 1),
    (2, 10, 64, 192, 128, 1, 1, 0),
    (2, 20, 128, 64, 128, 1, 2, 0),
    (1, 12, 200, 64, 128, 3, 2, 1),      # Wo = 100: the default dispatch takes this one
    (1, 10, 104, 192, 128, 1, 1, 0),
    # thin-input MFMA kernel (conv_thin_in.hip): stride-2 4x4 stems, 128-channel heads' input gradient
    (4, 66, 70, 3, 64, 4, 2, 1),
    (4, 66, 70, 1, 64, 4, 2, 1),
    (4, 60, 70, 128, 1, 3, 1, 1),
    # large enough maps for the thin-channel kernels (conv_thin.hip tile kernel, conv_wgrad_thin.hip)
    (4, 130, 140, 1, 16, 3, 1, 1),
    (4, 130, 140, 3, 48, 3, 1, 1),
    (4, 130, 140, 96, 1, 3, 1, 1),
    (2, 190, 180, 128, 1, 3, 1, 1),
    # 128- and 256-channel 3x3 layers (the LDS-DMA weight-gradient tiles; RDFC_WGRAD_GLDS=5 runs them on the 128 x 128 variant)
    (2, 20, 38, 128, 128, 3, 1, 1),
    (1, 29, 38, 256, 256, 3, 1, 1),
    (2, 12, 70, 128, 256, 3, 1, 0),         # >= 65536 pixels: the streaming head input-gradient kernel (conv_head_dgrad.hip)
    (4, 134, 146, 1, 64, 7, 1, 0),
    (4, 134, 146, 64, 3, 7, 1, 0),
    # round 4's wave-per-task thin-input kernel (conv_thin_in.hip) takes launches of >= 256 tasks of 4 x 16 output pixels:
    # PatchGAN's 4x4 / stride-2 first layers, the 3 -> 64 stem, an odd width
    (4, 130, 140, 3, 64, 4, 2, 1),
    (4, 130, 140, 1, 64, 4, 2, 1),
    (4, 130, 141, 3, 64, 3, 1, 1),
    # round 5: the 128 -> 32 head convs' weight gradient on the ring kernel's 64-channel tile (upper half beyond Cout: range-checked zeros)
    (4, 48, 64, 128, 32, 3, 1, 1),
    # round 6: the narrow-map kernels (conv_narrow.hip): NLSPN's conv_offset_aff (8 -> 24: forward, input gradient 24 -> 8, weight gradient)
    # and the input gradient of gd_dec0 (128 -> 8, gathered map of 8 channels -> 128); odd sizes, partial tiles
    (4, 66, 130, 8, 24, 3, 1, 1),
    (2, 35, 71, 128, 8, 3, 1, 1),
    (2, 21, 50, 24, 16, 3, 1, 1),
]


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(case, dtype):
    h = hip()
    N, H, W, Ci, Co, k, s, p = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    # the reference sees the same rounded operands as the kernel
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    xr = xq.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y_ref = F.leaky_relu(F.conv2d(xr, wr, b, s, p), 0.2)
    gy = rnd(*y_ref.shape, seed=4)
    gyq = gy.to(dtype).float()
    pre = F.conv2d(xr, wr, None, s, p)
    pre.backward(gyq)

    x_pm = to_pm(x, dtype, ld_extra=8)
    wf, wb = h.pack_weight(w.to(DEV), dtype)
    Ho, Wo = y_ref.shape[2:]
    y_pm = torch.empty(N, Ho, Wo, Co, dtype=dtype, device=DEV)
    d = h.conv_desc(x_pm.shape, Co, k, s, p, ldx=h.ld(x_pm), ldy=h.ld(y_pm), dtype=h.dt(x_pm), act='lrelu', slope=0.2)
    h.conv_forward(d, x_pm, wf, b.to(DEV), y_pm)
    assert rel_err(from_pm(y_pm), y_ref.detach()) < tol(dtype), 'forward'

    # f32 output flag
    y32 = torch.empty(N, Ho, Wo, Co, dtype=torch.float32, device=DEV)
    d32 = h.conv_desc(x_pm.shape, Co, k, s, p, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm), act='none', out_f32=True)
    h.conv_forward(d32, x_pm, wf, None, y32)
    assert rel_err(from_pm(y32), pre.detach()) < tol(dtype), 'forward f32 out'

    gy_pm = to_pm(gy, dtype)
    dx_pm = torch.empty(N, H, W, Ci, dtype=dtype, device=DEV)
    dd = h.conv_desc(x_pm.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=h.dt(x_pm))
    h.conv_dgrad(dd, gy_pm, wb, None, dx_pm)
    assert rel_err(from_pm(dx_pm), xr.grad) < tol(dtype), 'dgrad'

    dwd = h.conv_desc(x_pm.shape, Co, k, s, p, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm))
    dw = h.conv_wgrad(dwd, x_pm, gy_pm)
    gw = h.unpack_wgrad(dw, Co, Ci, k, k)
    assert rel_err(gw, wr.grad) < tol(dtype), 'wgrad'


NARROW_CASES = [
    # N, H, W, Ci, Co: what each direction must run on (conv_narrow.hip); None = another kernel's shape
    (4, 66, 130, 8, 24, 'narrow3x3_kernel<8, 2', 'narrow3x3_kernel<24, 1', 'false>'),       # conv_offset_aff (nlspn_model.py:29-33)
    (4, 66, 130, 128, 8, None, 'narrow3x3_kernel<8, 8', 'true>'),                            # gd_dec0 (rdf_generator.py:95-102): input and weight gradient
    (3, 41, 97, 128, 8, None, 'narrow3x3_kernel<8, 8', None),
    (2, 33, 70, 8, 8, 'narrow3x3_kernel<8, 1', 'narrow3x3_kernel<8, 1', None),
    (8, 70, 64, 8, 32, 'narrow3x3_kernel<8, 2', None, 'false>'),
    (9, 59, 64, 32, 8, None, 'narrow3x3_kernel<8, 2', 'true>'),
    (2, 33, 70, 8, 64, 'narrow3x3_kernel<8, 4', None, None),
    (2, 33, 70, 24, 32, 'narrow3x3_kernel<24, 2', None, None),
]


@pytest.mark.parametrize('dtype', DTYPES[1:], ids=DT_IDS[1:])
@pytest.mark.parametrize('case', NARROW_CASES, ids=lambda c: 'x'.join(map(str, c[:5])))
def test_narrow_map_convs(case, dtype):
    """conv_narrow.hip (round 6): 3x3 / stride-1 / pad-1 layers around 8- and 24-channel maps -- forward with bias into a 2-byte and into an
    f32 result (a channel slice of a wider map as input), input gradient, weight gradient (accumulating into a non-zero buffer, and bit
    for bit the same on a second run: its partial sums are added in a fixed order) -- against F.conv2d in fp32 on identically rounded
    operands, and each direction on the kernel it was written for (rdfc_last_kernel)."""
    h = hip()
    N, H, W, Ci, Co, k_fwd, k_dgrad, k_wgrad = case
    last = lambda: h.lib().rdfc_last_kernel().decode()
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    xr = x.to(dtype).float().requires_grad_(True)
    wr = w.to(dtype).float().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, 1, 1)
    gy = rnd(*y_ref.shape, seed=4)
    y_ref.backward(gy.to(dtype).float())
    x_pm = to_pm(x, dtype, ld_extra=8)
    wf, wb = h.pack_weight(w.to(DEV), dtype)
    for out_f32 in (False, True):
        y_pm = torch.empty(N, H, W, Co, dtype=torch.float32 if out_f32 else dtype, device=DEV)
        d = h.conv_desc(x_pm.shape, Co, 3, 1, 1, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm), act='none', out_f32=out_f32)
        h.conv_forward(d, x_pm, wf, b.to(DEV), y_pm)
        if k_fwd is not None:
            assert k_fwd in last(), last()
        assert rel_err(from_pm(y_pm), y_ref.detach()) < tol(dtype), f'forward (f32 result: {out_f32})'
    gy_pm = to_pm(gy, dtype)
    dx_pm = torch.empty(N, H, W, Ci, dtype=dtype, device=DEV)
    dd = h.conv_desc(x_pm.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=h.dt(x_pm))
    h.conv_dgrad(dd, gy_pm, wb, None, dx_pm)
    if k_dgrad is not None:
        assert k_dgrad in last(), last()
    assert rel_err(from_pm(dx_pm), xr.grad) < tol(dtype), 'dgrad'
    dwd = h.conv_desc(x_pm.shape, Co, 3, 1, 1, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm))
    base = torch.full((Co, 9, Ci), 0.5, device=DEV)
    runs = []
    for _ in range(2):
        buf = base.clone()
        h.conv_wgrad(dwd, x_pm, gy_pm, into=buf)
        runs.append(buf)
    if k_wgrad is not None:
        assert 'narrow_wgrad_kernel' in last() and last().endswith(k_wgrad), last()
        assert torch.equal(runs[0], runs[1]), 'the narrow weight gradient must repeat bit for bit'
    assert rel_err(h.unpack_wgrad(runs[0] - 0.5, Co, Ci, 3, 3), wr.grad) < tol(dtype), 'wgrad'


@pytest.mark.parametrize('dtype', DTYPES[1:], ids=DT_IDS[1:])
@pytest.mark.parametrize('case', [(4, 130, 140, 3, 64, 3, 'relu'), (4, 130, 141, 1, 16, 3, 'lrelu'), (3, 131, 150, 3, 48, 3, 'relu'), (4, 134, 146, 1, 64, 7, 'lrelu')],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_stem_backward_in_one_pass(case, dtype):
    """rdfc_conv2d_wgrad_act (round 6): weight and bias gradient of conv + bias + ReLU / LeakyReLU with the activation's backward folded into
    the staging of the cotangent -- against the three passes it replaces (rdfc_act_bwd, rdfc_channel_sum, rdfc_conv2d_wgrad: same 16-bit dy,
    so only the order of the f32 adds differs) and against autograd in fp32 on identically rounded operands; the cotangent is a channel
    slice of a wider map, both gradients ACCUMULATE into non-zero buffers."""
    h = hip()
    N, H, W, Ci, Co, k, act = case
    pad = k // 2
    slope = 0.2 if act == 'lrelu' else 0.0
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    xr, wr, br = x.to(dtype).float(), w.to(dtype).float().requires_grad_(True), b.clone().requires_grad_(True)
    x_pm = to_pm(x, dtype)
    wf, _ = h.pack_weight(w.to(DEV), dtype)
    y = torch.empty(N, H, W, Co, dtype=dtype, device=DEV)
    d = h.conv_desc(x_pm.shape, Co, k, 1, pad, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm), act=act, slope=slope)
    h.conv_forward(d, x_pm, wf, b.to(DEV), y)
    gz = rnd(N, Co, H, W, seed=4)
    dz = to_pm(gz, dtype, ld_extra=8)
    dd = h.conv_desc(x_pm.shape, Co, k, 1, pad, ldx=h.ld(x_pm), ldy=h.ld(dz), dtype=h.dt(x_pm), act=act, slope=slope)
    assert h.conv_wgrad_act_supported(dd)
    dw = torch.full((Co, k * k, Ci), 0.25, device=DEV)
    db = torch.full((Co,), -0.5, device=DEV)
    h.conv_wgrad_act(dd, x_pm, dz, y, dw, db)
    assert 'wgrad_thin_kernel' in h.lib().rdfc_last_kernel().decode() and 'true' in h.lib().rdfc_last_kernel().decode()
    # the three passes it replaces
    dy = h.act_bwd(dz, y, act, slope)
    dw3 = h.conv_wgrad(h.conv_desc(x_pm.shape, Co, k, 1, pad, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm)), x_pm, dy)
    db3 = h.channel_sum(dy)
    assert rel_err(dw - 0.25, dw3) < 2e-5 and rel_err(db + 0.5, db3) < 2e-5
    # autograd: the mask comes from the STORED (rounded) output, as in the kernels
    pre = F.conv2d(xr, wr, br, 1, pad)
    mask = from_pm(y) > 0
    g_pre = gz.to(dtype).float() * torch.where(mask, torch.ones(()), torch.full((), slope))
    pre.backward(g_pre.to(dtype).float())
    assert rel_err(h.unpack_wgrad(dw - 0.25, Co, Ci, k, k), wr.grad) < tol(dtype)
    assert rel_err(db + 0.5, br.grad) < tol(dtype)


DILATED_CASES = [
    # N, H, W, Ci, Co, k, s, p, dil   (label net: dilated ResNet-50 stages, backbone_resnet50.py:126-235)
    (2, 15, 19, 64, 64, 3, 1, 2, 2),
    (1, 29, 38, 128, 128, 3, 1, 4, 4),
    (2, 17, 21, 256, 128, 3, 1, 8, 8),
    (1, 29, 38, 64, 64, 3, 1, 16, 16),
    (2, 15, 19, 128, 64, 3, 2, 2, 2),
]


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', DILATED_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_dilated(case, dtype):
    h = hip()
    N, H, W, Ci, Co, k, s, p, dil = case
    x = rnd(N, Ci, H, W, seed=1).to(dtype).float()
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5).to(dtype).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p, dil)
    gy = rnd(*y_ref.shape, seed=4).to(dtype).float()
    y_ref.backward(gy)
    x_pm, gy_pm = to_pm(x, dtype), to_pm(gy, dtype)
    wf, wb = h.pack_weight(w.to(DEV), dtype)
    Ho, Wo = y_ref.shape[2:]
    y_pm = torch.empty(N, Ho, Wo, Co, dtype=dtype, device=DEV)
    d = h.conv_desc(x_pm.shape, Co, k, s, p, dil, ldx=Ci, ldy=Co, dtype=h.dt(x_pm))
    assert (d.Ho, d.Wo) == (Ho, Wo)
    h.conv_forward(d, x_pm, wf, None, y_pm)
    assert rel_err(from_pm(y_pm), y_ref.detach()) < tol(dtype), 'forward'
    dx_pm = torch.empty(N, H, W, Ci, dtype=dtype, device=DEV)
    h.conv_dgrad(d, gy_pm, wb, None, dx_pm)
    assert rel_err(from_pm(dx_pm), xr.grad) < tol(dtype), 'dgrad'
    gw = h.unpack_wgrad(h.conv_wgrad(d, x_pm, gy_pm), Co, Ci, k, k)
    assert rel_err(gw, wr.grad) < tol(dtype), 'wgrad'


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', [(2, 5, 7, 512, 256), (2, 8, 10, 192, 64), (1, 15, 19, 128, 64)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_conv_transpose_through_primitives(case, dtype):
    """ConvTranspose2d(3, s2, p1, op1): forward = dgrad primitive, input-grad = forward primitive,
    weight-grad = wgrad primitive with x <-> dy."""
    h = hip()
    N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=1).to(dtype).float()
    w = (rnd(Cin, Cout, 3, 3, seed=2) * (Cin * 9) ** -0.5).to(dtype).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv_transpose2d(xr, wr, None, 2, 1, 1)
    gy = rnd(*y_ref.shape, seed=3).to(dtype).float()
    y_ref.backward(gy)
    # equivalent conv: x_c = Y_T space (Ci_c = Cout), y_c = X_T space (Co_c = Cin), weight [O=Cin][I=Cout]
    wf, wb = h.pack_weight(w.to(DEV), dtype)
    x_pm = to_pm(x, dtype)
    y_pm = torch.empty(N, 2 * H, 2 * W, Cout, dtype=dtype, device=DEV)
    d = h.conv_desc((N, 2 * H, 2 * W, Cout), Cin, 3, 2, 1, ldx=Cout, ldy=Cin, dtype=h.dt(x_pm), out_hw=(H, W))
    h.conv_dgrad(d, x_pm, wb, None, y_pm)
    assert rel_err(from_pm(y_pm), y_ref.detach()) < tol(dtype), 'convT forward'
    gy_pm = to_pm(gy, dtype)
    dx_pm = torch.empty(N, H, W, Cin, dtype=dtype, device=DEV)
    h.conv_forward(d, gy_pm, wf, None, dx_pm)
    assert rel_err(from_pm(dx_pm), xr.grad) < tol(dtype), 'convT input grad'
    dw = h.conv_wgrad(d, gy_pm, x_pm)
    assert rel_err(h.unpack_wgrad(dw, Cin, Cout, 3, 3), wr.grad) < tol(dtype), 'convT weight grad'


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('mode', ['bn', 'in'])
@pytest.mark.parametrize('act,res', [('relu', True), ('lrelu', False), ('prelu', False), ('none', False)])
def test_norm_fwd_bwd(dtype, mode, act, res):
    h = hip()
    N, H, W, C = 3, 9, 11, 96
    x = (rnd(N, C, H, W, seed=1) * 1.5 + 0.3).to(dtype).float()
    r = rnd(N, C, H, W, seed=2).to(dtype).float() if res else None
    gamma = (1 + 0.1 * rnd(C, seed=3)) if mode == 'bn' else None
    beta = 0.1 * rnd(C, seed=4) if mode == 'bn' else None
    slope_t = torch.tensor([0.25])
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    gr = gamma.clone().requires_grad_(True) if gamma is not None else None
    br = beta.clone().requires_grad_(True) if beta is not None else None
    ar = slope_t.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    if mode == 'bn':
        u = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    else:
        u = F.instance_norm(xr, eps=1e-5)
    if res:
        u = u + rr
    z_ref = {'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.2), 'prelu': lambda t: F.prelu(t, ar),
             'none': lambda t: t}[act](u)
    gz = rnd(N, C, H, W, seed=5).to(dtype).float()
    z_ref.backward(gz)

    x_pm = to_pm(x, dtype, ld_extra=8)
    r_pm = to_pm(r, dtype) if res else None
    G = 1 if mode == 'bn' else N
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, rstd = h.norm_stats(x_pm, G, 1e-5, rm_d if mode == 'bn' else None, rv_d if mode == 'bn' else None, 0.1)
    z = torch.empty(N, H, W, C, dtype=dtype, device=DEV)
    gd = gamma.to(DEV) if gamma is not None else None
    bd = beta.to(DEV) if beta is not None else None
    pr = slope_t.to(DEV)
    h.norm_apply(x_pm, mean, rstd, gd, bd, r_pm, z, act, 0.2, pr)
    assert rel_err(from_pm(z), z_ref.detach()) < tol(dtype), 'forward'
    if mode == 'bn':
        assert rel_err(rm_d, rm) < 1e-4 and rel_err(rv_d, rv) < 1e-4, 'running stats'
    gz_pm = to_pm(gz, dtype)
    dx, dres, sums = h.norm_backward(gz_pm, x_pm, mean, rstd, gd, bd, r_pm, act, 0.2, pr, want_dres=res)
    t = 4 * tol(dtype)
    assert rel_err(from_pm(dx), xr.grad) < t, 'dx'
    if res:
        assert rel_err(from_pm(dres), rr.grad) < t, 'dres'
    if mode == 'bn':
        assert rel_err(sums[0, 1], gr.grad) < t, 'dgamma'
        assert rel_err(sums[0, 0], br.grad) < t, 'dbeta'
    if act == 'prelu':
        assert rel_err(sums[:, 2].sum(), ar.grad.sum()) < t, 'dslope'


@pytest.mark.parametrize('case', [(3, 21, 37, 64, 64, 1), (2, 19, 23, 64, 128, 0), (4, 18, 20, 96, 256, 1)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_conv_fused_statistics(case, monkeypatch):
    """rdfc_conv2d_forward_stats (default since round 4, RDFC_FUSED_STATS=0 switches it off): the conv epilogue's per-tile statistics +
    rdfc_norm_finalize_n against the separate rdfc_norm_stats pass over the stored map, batch-norm (G = 1, with running statistics) and
    instance-norm (G = N); then BatchNorm over two stacked batches through norm_from_partials (statistics per group in parallel, running
    statistics advanced by the normalise pass) against norm_forward(group_momenta) on the stored map."""
    h = hip()
    N, H, W, Ci, Co, pad = case
    dtype = torch.bfloat16
    if h.conv_stats_nchunk(h.conv_desc((N, H, W, Ci), Co, 3, 1, pad, ldx=Ci, ldy=Co, dtype=h.dt(torch.empty(0, dtype=dtype))), 1) == 0:
        pytest.skip('fused statistics switched off (RDFC_FUSED_STATS=0)')
    x = to_pm(rnd(N, Ci, H, W, seed=1), dtype)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5).to(DEV)
    b = rnd(Co, seed=3, scale=0.1).to(DEV)
    wf, _ = h.pack_weight(w, dtype)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    d = h.conv_desc(x.shape, Co, 3, 1, pad, ldx=h.ld(x), ldy=Co, dtype=h.dt(x))
    for G in (1, N):
        assert h.conv_stats_nchunk(d, G) > 0
        y_ref = torch.empty(N, Ho, Wo, Co, dtype=dtype, device=DEV)
        h.conv_forward(d, x, wf, b, y_ref)
        rm1, rv1 = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        rm2, rv2 = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        upd = G == 1
        mean_ref, rstd_ref = h.norm_stats(y_ref, G, 1e-5, rm1 if upd else None, rv1 if upd else None, 0.1)
        y = torch.empty_like(y_ref)
        mean, rstd = h.conv_forward_stats(d, x, wf, b, y, G, 1e-5, rm2 if upd else None, rv2 if upd else None, 0.1)
        assert torch.equal(y, y_ref)
        assert rel_err(mean, mean_ref) < 2e-5 and rel_err(rstd, rstd_ref) < 2e-5
        if upd:
            assert rel_err(rm2, rm1) < 2e-5 and rel_err(rv2, rv1) < 2e-5
    if N % 2 == 0:                          # two stacked batches: the launch tiles the batch group by group (no tile straddles the two)
        moms = (0.1, 0.19)
        gamma, beta = rnd(Co, seed=5).to(DEV) + 1.5, rnd(Co, seed=6).to(DEV)
        y_ref = torch.empty(N, Ho, Wo, Co, dtype=dtype, device=DEV)
        h.conv_forward(d, x, wf, b, y_ref)
        rm1, rv1 = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        rm2, rv2 = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        z_ref, z = torch.empty_like(y_ref), torch.empty_like(y_ref)
        mean_ref, rstd_ref = h.norm_forward(y_ref, 2, 1e-5, gamma, beta, None, z_ref, 'lrelu', 0.2, None, rm1, rv1, 0.1, group_momenta=moms)
        y = torch.empty_like(y_ref)
        buf, n = h.conv_forward_partials(d, x, wf, b, y, 2)
        mean, rstd = h.norm_from_partials(buf, n, y, 2, 1e-5, gamma, beta, None, z, 'lrelu', 0.2, None, rm2, rv2, moms)
        assert torch.equal(y, y_ref)
        assert rel_err(mean, mean_ref) < 2e-5 and rel_err(rstd, rstd_ref) < 2e-5
        assert rel_err(z, z_ref) < 1e-2 and rel_err(rm2, rm1) < 2e-5 and rel_err(rv2, rv1) < 2e-5


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('shape,G', [((4, 57, 76, 256), 4), ((2, 60, 76, 64), 1), ((16, 19, 23, 128), 16), ((3, 31, 33, 40), 3)],
                         ids=['in256', 'bn64', 'in128x16', 'in40'])
def test_bias_gradient_from_the_norm_backward_apply_pass(shape, G, dtype):
    """round 6 (rdfc_norm_backward_dxsum): the gradient of a conv bias in front of a norm layer = sum of dx over the pixels, taken by the
    apply pass while it stores dx.  Must be the sum of the STORED values (what rdfc_channel_sum_acc over dx returns, up to the order of the
    f32 adds of ~1e5 rounding errors: compared against the sum of |dx|), accumulate into its buffer, leave dx / the sums untouched and
    repeat bit for bit (fixed-order two-stage sum)."""
    h = hip()
    N, H, W, C = shape
    x = to_pm(rnd(N, C, H, W, seed=1), dtype)
    dz = to_pm(rnd(N, C, H, W, seed=2), dtype, ld_extra=8)
    P = N * H * W // G
    mean, rstd = h.norm_stats(x, G, 1e-5)
    gamma = (rnd(C, seed=3).to(DEV) * 0.1 + 1.0) if G == 1 else None
    beta = (rnd(C, seed=4).to(DEV) * 0.1) if G == 1 else None
    dx0, _, sums0 = h.norm_backward(dz, x, mean, rstd, gamma, beta, None, 'relu', 0.0, None)
    ref = h.channel_sum(dx0)
    runs = []
    for _ in range(2):
        acc = torch.full((C,), 3.0, device=DEV)
        dx1, _, sums1 = h.norm_backward(dz, x, mean, rstd, gamma, beta, None, 'relu', 0.0, None, acc_bias=acc)
        assert torch.equal(dx1, dx0) and torch.equal(sums1, sums0)
        runs.append(acc)
    assert torch.equal(runs[0], runs[1])
    scale = float(dx0.float().abs().sum(dim=(0, 1, 2)).max())
    assert float((runs[0] - 3.0 - ref).abs().max()) <= 2e-6 * scale + 1e-6, (float((runs[0] - 3.0 - ref).abs().max()), scale)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_adain(dtype):
    h = hip()
    N, H, W, C = 2, 7, 9, 192
    x = rnd(N, C, H, W, seed=1).to(dtype).float()
    gb = (rnd(N, 2 * C, H, W, seed=2) * 0.5).to(dtype).float()
    xr, gbr = x.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    out_ref = gbr[:, :C] * F.instance_norm(xr, eps=1e-5) + gbr[:, C:]
    go = rnd(N, C, H, W, seed=3).to(dtype).float()
    out_ref.backward(go)
    x_pm, gb_pm = to_pm(x, dtype), to_pm(gb, dtype)
    mean, rstd = h.norm_stats(x_pm, N, 1e-5)
    out = torch.empty(N, H, W, C, dtype=dtype, device=DEV)
    h.adain_apply(x_pm, mean, rstd, gb_pm, out)
    assert rel_err(from_pm(out), out_ref.detach()) < tol(dtype)
    dx, dgb = h.adain_backward(to_pm(go, dtype), x_pm, mean, rstd, gb_pm)
    assert rel_err(from_pm(dx), xr.grad) < 4 * tol(dtype)
    assert rel_err(from_pm(dgb), gbr.grad) < 4 * tol(dtype)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_layout_window_pad_sum(dtype):
    h = hip()
    x = rnd(2, 24, 9, 7, seed=1).to(dtype).float()
    pm = h.nchw_to_pm(x.to(DEV), dtype)
    assert rel_err(from_pm(pm), x) == 0
    assert rel_err(h.pm_to_nchw(pm), x) == 0
    x3 = rnd(2, 3, 9, 7, seed=2)
    assert rel_err(h.pm_to_nchw(h.nchw_to_pm(x3.to(DEV), torch.float32)), x3) == 0
    # crop + channel-slice copy (torch.cat of a cropped decoder map and a skip map)
    dst = torch.zeros(2, 8, 6, 48, dtype=dtype, device=DEV)
    h.copy_window(pm, dst[..., :24])
    h.copy_window(pm[:, :8, :6].contiguous(), dst[..., 24:])
    ref = torch.cat([x[:, :, :8, :6], x[:, :, :8, :6]], 1)
    assert rel_err(from_pm(dst), ref) == 0
    h.copy_window(pm, dst[..., :24], accumulate=True)
    assert rel_err(from_pm(dst[..., :24].contiguous()), 2 * x[:, :, :8, :6]) < tol(dtype)
    # reflection pad and its adjoint
    for pad in (1, 3):
        xr = x.clone().requires_grad_(True)
        yp = F.pad(xr, (pad,) * 4, mode='reflect')
        g = rnd(*yp.shape, seed=3).to(dtype).float()
        yp.backward(g)
        assert rel_err(from_pm(h.reflect_pad(pm, pad)), yp.detach()) == 0
        assert rel_err(from_pm(h.reflect_pad_adjoint(to_pm(g, dtype), pad)), xr.grad) < tol(dtype)
    # per-channel sums, odd channel counts
    for C in (1, 3, 24, 1536 if dtype == torch.float32 else 768):
        t = rnd(2, C, 5, 6, seed=C).to(dtype).float()
        assert rel_err(h.channel_sum(to_pm(t, dtype)), t.sum((0, 2, 3))) < tol(dtype)
    # activation backward from the output
    y = torch.tanh(rnd(2, 1, 9, 7, seed=5))
    g = rnd(2, 1, 9, 7, seed=6)
    got = h.act_bwd(to_pm(g, torch.float32), to_pm(y, torch.float32), 'tanh')
    assert rel_err(from_pm(got), g * (1 - y * y)) < 1e-6


def test_fusion_head_losses_adam_pepper():
    h = hip()
    shp = (2, 1, 9, 11)
    d1, c1, d2, c2 = (rnd(*shp, seed=i) * (1.5 if i == 3 else 1.0) for i in range(1, 5))
    leaves = [t.clone().requires_grad_(True) for t in (d1, c1, d2, c2)]
    d2c = leaves[2].clamp(-1, 1)
    score = torch.softmax(torch.cat([leaves[1], leaves[3]], 1), 1)
    pred = (torch.cat([leaves[0], d2c], 1) * score).sum(1, keepdim=True)
    gp, gd = rnd(*shp, seed=7), rnd(*shp, seed=8)
    (pred * gp).sum().backward(retain_graph=True)
    (d2c * gd).sum().backward()
    dev = [t.to(DEV) for t in (d1, c1, d2, c2)]
    o_d2c, o_pred = h.fusion_head(*dev)
    assert rel_err(o_pred, pred.detach()) < 1e-6 and rel_err(o_d2c, d2c.detach()) == 0
    grads = h.fusion_head_bwd(*dev, gp.to(DEV), gd.to(DEV))
    for g, leaf in zip(grads, leaves):
        assert rel_err(g, leaf.grad) < 1e-5
    # losses
    for C in (1, 3):
        a, b = rnd(2, C, 20, 24, seed=1), rnd(2, C, 20, 24, seed=2)
        m = torch.rand(2, 1, 20, 24, generator=torch.Generator().manual_seed(3)) > 0.3
        w = m / (m.sum() + 1e-6)
        for kind, fn in ((1, lambda p, t: (p - t).abs()), (2, lambda p, t: (p - t) ** 2)):
            for ww in (w, None):
                ar = a.clone().requires_grad_(True)
                wt = ww if ww is not None else torch.ones_like(a) / (a.numel() + 1e-6)
                ref = (wt * fn(ar, b)).sum()
                (ref * 3.0).backward()
                val, grad = h.weighted_loss(a.to(DEV), b.to(DEV), ww.to(DEV) if ww is not None else None, kind, True, 3.0)
                assert abs(float(val) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
                assert rel_err(grad, ar.grad) < 1e-5
    p = rnd(2, 1, 26, 36, seed=5)
    val, grad = h.weighted_loss(p.to(DEV), 1.0, None, 2, True, 1.0)
    ref = ((p - 1) ** 2).sum() / (p.numel() + 1e-6)
    assert abs(float(val) - float(ref)) < 1e-5
    # Adam against torch.optim.Adam
    prm = torch.nn.Parameter(rnd(1000, seed=1))
    opt = torch.optim.Adam([prm], lr=2e-3, betas=(0.5, 0.999))
    pd, md, vd = prm.detach().clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 4):
        g = rnd(1000, seed=10 + step)
        prm.grad = g.clone()
        opt.step()
        h.adam_step(pd, g.to(DEV), md, vd, 2e-3, 0.5, 0.999, 1e-8, step)
    assert rel_err(pd, prm.detach()) < 1e-6
    # pepper
    img = rnd(2, 3, 5, 6, seed=1)
    mask = torch.randint(0, 3, (2, 1, 5, 6), generator=torch.Generator().manual_seed(2), dtype=torch.uint8)
    ref = img.clone()
    mm = mask.expand(2, 3, 5, 6)
    ref[mm == 1] = 1
    ref[mm == 2] = -1
    got = h.apply_pepper(img.to(DEV), mask.to(DEV))
    assert rel_err(got, ref) == 0


def test_dynamic_loss_scale_skips_overflow_and_backs_off():
    """rdfc_grad_nonfinite / rdfc_adam_step_scaled / rdfc_loss_scale_update (f16 mode's device-side scaler): a clean step equals
    torch.optim.Adam on the unscaled gradients; a step whose (scaled) gradients hold an inf / NaN changes nothing, is counted and
    halves the scale; the bias corrections of later steps count the steps really taken; growth after `interval` clean steps."""
    h = hip()
    n = 1003                                        # (not a multiple of 4: the check's scalar tail)
    prm = torch.nn.Parameter(rnd(n, seed=1))
    opt = torch.optim.AdamW([prm], lr=2e-3, betas=(0.5, 0.999), weight_decay=0.01)
    pd, md, vd = prm.detach().clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    scaler = torch.tensor([1024.0, 1.0 / 1024.0, 0.0, 0.0], device=DEV)
    ost = torch.zeros(2, device=DEV)
    host_step = 0
    for it, bad in enumerate([None, 'inf', None, 'nan', None]):
        g = rnd(n, seed=10 + it)
        gs = (g * float(scaler[0])).to(DEV)          # what a backward started from the cotangent `scale` leaves in .grad
        if bad is not None:
            gs[n - 1 if bad == 'inf' else 17] = float(bad)
        before = (pd.clone(), md.clone(), vd.clone())
        host_step += 1
        h.grad_nonfinite(gs, ost)
        h.adam_step_scaled(pd, gs, md, vd, 2e-3, 0.5, 0.999, 1e-8, host_step, scaler, ost, grad_scale=1.0, weight_decay=0.01)
        h.loss_scale_update(scaler, 2.0, 0.5, 2, 1.0, 2.0 ** 24)
        if bad is None:
            prm.grad = g.clone()
            opt.step()
        else:
            assert all(torch.equal(a, b) for a, b in zip(before, (pd, md, vd))), 'an overflowing step must not touch the state'
        assert rel_err(pd, prm.detach()) < 2e-6, (it, bad)
    # skipped twice; scale: 1024 -> (clean, 1 of 2) -> 512 (inf) -> (clean) -> 256 (nan) -> (clean, 1 of 2) = 256
    assert float(ost[1]) == 2.0 and float(scaler[0]) == 256.0 and abs(float(scaler[1]) - 1.0 / 256.0) < 1e-9
    h.grad_nonfinite(torch.zeros(8, device=DEV), ost)
    h.loss_scale_update(scaler, 2.0, 0.5, 2, 1.0, 2.0 ** 24)             # second clean step in a row: grows
    assert float(scaler[0]) == 512.0 and float(scaler[2]) == 0.0 and float(ost[0]) == 0.0


MD_CASES = [dict(B=2, C=1, H=9, W=11, Co=1, k=3, s=1, p=1, g=1, dg=1),
            dict(B=2, C=1, H=21, W=150, Co=1, k=3, s=1, p=1, g=1, dg=1),      # several 8x64 tiles of the LDS-window backward
            dict(B=1, C=1, H=19, W=70, Co=1, k=1, s=1, p=0, g=1, dg=1),
            dict(B=2, C=1, H=9, W=11, Co=1, k=1, s=1, p=0, g=1, dg=1),
            dict(B=2, C=4, H=7, W=9, Co=6, k=3, s=1, p=1, g=2, dg=2),
            dict(B=1, C=4, H=8, W=8, Co=4, k=3, s=2, p=1, g=1, dg=1)]


@pytest.mark.parametrize('c', MD_CASES, ids=lambda c: f"C{c['C']}k{c['k']}s{c['s']}g{c['g']}W{c['W']}")
def test_mdconv_against_oracle(c):
    """rdfc_mdconv_* against the C oracle (which restates upstream's CUDA kernels)."""
    from oracle import mdconv as om
    h = hip()
    k, s, p = c['k'], c['s'], c['p']
    Ho = (c['H'] + 2 * p - k) // s + 1
    Wo = (c['W'] + 2 * p - k) // s + 1
    x = rnd(c['B'], c['C'], c['H'], c['W'], seed=1)
    off = rnd(c['B'], c['dg'] * 2 * k * k, Ho, Wo, seed=2, scale=2.0)
    msk = torch.sigmoid(rnd(c['B'], c['dg'] * k * k, Ho, Wo, seed=3)) * 2
    w = rnd(c['Co'], c['C'] // c['g'], k, k, seed=4, scale=0.3)
    b = rnd(c['Co'], seed=5)
    go = rnd(c['B'], c['Co'], Ho, Wo, seed=6)
    ref = om.forward(x, off, msk, w, b, s, p, 1, c['g'], c['dg'])
    refs = om.backward(x, off, msk, w, go, s, p, 1, c['g'], c['dg'])
    dv = [t.to(DEV) for t in (x, off, msk, w)]
    got = h.mdconv_forward(*dv, b.to(DEV), (s, s), (p, p), (1, 1), c['g'], c['dg'])
    assert rel_err(got, ref) < 1e-5
    gots = h.mdconv_backward(*dv, go.to(DEV), (s, s), (p, p), (1, 1), c['g'], c['dg'])
    for name, a, r in zip(('input', 'offset', 'mask', 'weight', 'bias'), gots, refs):
        assert rel_err(a, r) < 2e-5, name


def test_depth_metrics_against_oracle_and_golden(golden):
    """rdfc_depth_metrics / rdfc_gan_amd.metrics.RDFGANMetric against the CPU oracle and the reference's own numbers"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from cases import metric_case
    from oracle import metrics as om
    from rdfc_gan_amd.metrics import RDFGANMetric
    g = golden('metric')
    gt_n, pd_n, ev = metric_case()
    m = RDFGANMetric()
    per = m.evaluate_normalised(gt_n.to(DEV), pd_n.to(DEV), ev.to(DEV)).double().cpu().numpy()
    np.testing.assert_allclose(per, g['per_image'], rtol=2e-5, atol=1e-6)
    real, fake = om.denormalise(gt_n.numpy()), om.denormalise(pd_n.numpy())
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        plain = m.evaluate_all([dict(gt=real[k].squeeze(), pd=fake[k].squeeze()) for k in range(len(real))])
    np.testing.assert_allclose([plain[n] for n in m.metric_name], g['plain'], rtol=2e-5, atol=1e-6)
    # full NYUv2-sized batch: HIP against the oracle
    gt2, pd2, ev2 = metric_case(B=8, H=228, W=304, seed=5)
    got = m.evaluate_normalised(gt2.to(DEV), pd2.to(DEV), None).double().cpu().numpy()
    ref = np.stack([om.image_metrics(om.denormalise(gt2[k].numpy()), om.denormalise(pd2[k].numpy()), acc=np.float64) for k in range(8)])
    np.testing.assert_allclose(got, ref, rtol=5e-5, atol=1e-6)


def test_device_pepper_mask_distribution_and_determinism():
    """rdfc_draw_pepper_mask: the distribution of helper.py:384-402's np.random.choice((0, 1, 2), p=[snr, (1-snr)/2, (1-snr)/2]),
    stateless in (seed, counter, pixel); RDFCGAN(device_pepper=True) uses it instead of the host draw"""
    h = hip()
    a = h.draw_pepper_mask(8, 228, 304, 1234, 1, 0.98, DEV)
    b = h.draw_pepper_mask(8, 228, 304, 1234, 1, 0.98, DEV)
    c = h.draw_pepper_mask(8, 228, 304, 1234, 2, 0.98, DEV)
    d = h.draw_pepper_mask(8, 228, 304, 1235, 1, 0.98, DEV)
    assert a.shape == (8, 1, 228, 304) and a.dtype == torch.uint8
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    n = a.numel()
    f = [float((a == k).sum()) / n for k in (0, 1, 2)]
    sig = (0.01 * 0.99 / n) ** 0.5
    assert abs(f[0] - 0.98) < 6 * (0.98 * 0.02 / n) ** 0.5 and abs(f[1] - 0.01) < 6 * sig and abs(f[2] - 0.01) < 6 * sig, f
    # no structure along rows / between images: per-image replaced fractions scatter like a binomial
    per = (a != 0).float().mean(dim=(1, 2, 3)).cpu().numpy()
    assert np.abs(per - 0.02).max() < 6 * (0.02 * 0.98 / (228 * 304)) ** 0.5
    img = torch.zeros(8, 3, 228, 304, device=DEV)
    out = h.apply_pepper(img.clone(), a)
    assert torch.equal(out[:, 0], out[:, 1]) and torch.equal((out[:, 0] == 1), (a[:, 0] == 1)) and torch.equal((out[:, 0] == -1), (a[:, 0] == 2))


# ---- round 2: stacked-batch BatchNorm, fan-out sums, small elementwise operators ---------------------------------------
@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('chan', [(16, 32), (64, 64)], ids=['16to32', '64to64'])       # 64 -> 64: the LDS-DMA kernel (fused statistics under RDFC_FUSED_STATS=1)
@pytest.mark.parametrize('repeats', [(1, 2), (1, 1), (1, 1, 2)], ids=str)
def test_bn_over_stacked_batches_equals_separate_calls(dtype, repeats, chan):
    """conv + BatchNorm(train) + LeakyReLU over G stacked batches (layers.bn_groups) == G separate calls of the block:
    outputs, running statistics (advanced once per group, in order, group g standing for repeats[g] identical forwards),
    num_batches_tracked, input / weight / affine gradients."""
    import torch.nn as nn
    from rdfc_gan_amd import functional as Fh, layers
    Fh.set_compute_dtype(dtype)
    torch.manual_seed(0)
    G, n, (Ci, Co), H, W = len(repeats), 2, chan, 11, 13

    def make():
        torch.manual_seed(1)
        seq = layers.conv_bn_relu(Ci, Co, 3, 1, 1).to(DEV)
        with torch.no_grad():
            seq[1].weight.uniform_(0.5, 1.5)
            seq[1].bias.uniform_(-0.5, 0.5)
        return seq

    xs = [rnd(n, Ci, H, W, seed=10 + g, scale=1.0 + 0.5 * g).to(DEV) for g in range(G)]
    gys = [rnd(n, H, W, Co, seed=20 + g).to(DEV).to(dtype) for g in range(G)]
    # separate calls
    ref = make()
    ref.train()
    outs, gx_ref = [], []
    for g in range(G):
        x = Fh.to_pm(xs[g]).detach().requires_grad_(True)
        layers.BN_STAT_REPEAT = repeats[g]
        try:
            y = ref(x)
        finally:
            layers.BN_STAT_REPEAT = 1
        y.backward(gys[g])
        outs.append(y.detach().float())
        gx_ref.append(x.grad.float())
    # one stacked call
    got = make()
    got.train()
    x = Fh.to_pm(torch.cat(xs, 0)).detach().requires_grad_(True)
    with layers.bn_groups(repeats):
        y = got(x)
    y.backward(torch.cat(gys, 0))
    t = tol(dtype)
    assert rel_err(y.detach().float(), torch.cat(outs, 0)) < t
    assert rel_err(x.grad.float(), torch.cat(gx_ref, 0)) < 4 * t
    sd_ref, sd_got = ref.state_dict(), got.state_dict()
    assert int(sd_got['1.num_batches_tracked']) == int(sd_ref['1.num_batches_tracked']) == sum(repeats)
    for k in ('1.running_mean', '1.running_var'):
        assert rel_err(sd_got[k], sd_ref[k]) < 1e-6, k
    for name in ('0.weight', '1.weight', '1.bias'):
        a, b = dict(got.named_parameters())[name].grad, dict(ref.named_parameters())[name].grad
        assert rel_err(a, b) < 4 * t, name
    Fh.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_sum_maps_and_fanout(dtype):
    h = hip()
    from rdfc_gan_amd import functional as Fh
    N, H, W, C = 2, 9, 11, 24
    maps = [to_pm(rnd(N, C, H, W, seed=s), dtype, ld_extra=(8 if s % 2 else 0)) for s in range(4)]
    for n in (2, 3, 4):
        got = h.sum_maps(maps[:n])
        ref = sum(m.float() for m in maps[:n])
        assert rel_err(got.float(), ref) < {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    odd = [to_pm(rnd(N, 5, H, W, seed=s), dtype) for s in range(2)]         # channel count without a 16-byte vector
    assert rel_err(h.sum_maps(odd).float(), odd[0].float() + odd[1].float()) < {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    x = maps[0].contiguous().clone().requires_grad_(True)
    a, b, c = Fh.fanout(x, 3)
    (a.float() * 2.0).sum().backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    ((a.float() * 2.0).sum() + (b.float() * 3.0).sum() + (c[..., :8].float() * 5.0).sum()).backward()
    exp = torch.full_like(x, 5.0)
    exp[..., :8] += 5.0
    assert torch.equal(g1, torch.full_like(x, 2.0)) and torch.equal(x.grad, exp)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_chained_fanout_sums_inside_the_input_gradient_launches(dtype):
    """fanout(..., chained=True): conv-block consumers add the running sum of the cotangents in their own input-gradient epilogue.
    Same gradients as the rdfc_sum_maps form, no rdfc_sum_maps launch; consumers on two streams; an unused consumer; a consumer
    the chain does not cover (a channel slice)."""
    h = hip()
    from rdfc_gan_amd import functional as Fh
    N, H, W, C = 2, 20, 24, 32
    x0 = to_pm(rnd(N, C, H, W, seed=1), dtype).contiguous()
    ws = [rnd(co, C, 3, 3, seed=10 + i, scale=0.2).to(DEV) for i, co in enumerate((16, 32, 8))]
    spec = Fh.ConvSpec(3, 1, 1, act='lrelu', slope=0.2)
    side = torch.cuda.Stream()

    def run(chained, use=(0, 1, 2), sliced=False, two_streams=False):
        x = x0.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        outs = Fh.fanout(x, 3 + int(sliced), chained=chained)
        assert (getattr(outs[0], '_rdfc_chain', None) is not None) == chained
        total = 0.0
        for i in use:
            if two_streams and i == 1:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    y = Fh.conv_block(outs[i], wl[i], spec=spec)
                    total = total + (y.float() * (i + 1.5)).sum()
                torch.cuda.current_stream().wait_stream(side)
            else:
                y = Fh.conv_block(outs[i], wl[i], spec=spec)
                total = total + (y.float() * (i + 1.5)).sum()
        if sliced:
            total = total + (outs[3][..., :8].float() * 0.75).sum()
        calls = []
        orig = h.sum_maps
        h.sum_maps = lambda maps: (calls.append(len(maps)), orig(maps))[1]
        try:
            total.backward()
        finally:
            h.sum_maps = orig
        torch.cuda.synchronize()
        return x.grad.float(), [w.grad for w in wl], calls

    t = {torch.float32: 2e-6, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}[dtype]
    Fh.set_compute_dtype(dtype)
    try:
        for kw in (dict(), dict(two_streams=True), dict(use=(0, 2)), dict(sliced=True), dict(use=(1,))):
            g_ref, w_ref, c_ref = run(False, **kw)
            g, wg, calls = run(True, **kw)
            assert rel_err(g, g_ref) < t, kw
            for a, b in zip(wg, w_ref):
                assert (a is None) == (b is None) and (a is None or rel_err(a, b) < 1e-5)       # (same launches; f32 atomics in some)
            n_in = len(kw.get('use', (0, 1, 2))) + int(kw.get('sliced', False))
            assert c_ref == ([n_in] if n_in > 1 else [])
            # the chain leaves one rdfc_sum_maps only where a cotangent arrives from outside it
            assert calls == ([2] if kw.get('sliced') else []), (kw, calls)
    finally:
        Fh.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('order', ['skip_first', 'stage_first'])
def test_chained_fanout_skip_connection_joins_the_next_stage_first_input_gradient(dtype, order):
    """The encoder's pattern: an activation feeds the next stage's first block (3x3 / stride 2, which also hands out an alias of its input
    to the 1x1 / stride-2 downsample path) and a decoder skip connection (the channel slice of a concatenation's cotangent).  Chained, the
    slice is deposited and the stride-2 input-gradient launch adds it beside the alias cotangent (rdfc_conv2d_dgrad_add2): same gradients,
    no rdfc_sum_maps; with the stage's backward first the deposit is summed by _FanOut.backward instead."""
    h = hip()
    from rdfc_gan_amd import functional as Fh
    N, H, W, C = 16, 48, 64, 64           # (large enough for the transposed stride-2 kernel to take the launch)
    x0 = to_pm(rnd(N, C, H, W, seed=3), dtype).contiguous()
    w1 = rnd(128, C, 3, 3, seed=21, scale=0.1).to(DEV)
    wd = rnd(128, C, 1, 1, seed=22, scale=0.3).to(DEV)
    wc = rnd(32, 16, 3, 3, seed=23, scale=0.2).to(DEV)
    y0 = to_pm(rnd(N, 16, H, W, seed=4), dtype).contiguous()
    s1 = Fh.ConvSpec(3, 2, 1, act='relu', passthrough=True)
    sd = Fh.ConvSpec(1, 2, 0)
    sc = Fh.ConvSpec(3, 1, 1, norm='bn', act='relu')
    gam, bet = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)

    def run(chained, via_crop_cat=False):
        x = x0.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in (w1, wd, wc)]
        nxt, skip = Fh.fanout(x, 2, chained=chained)

        def stage():
            y, xr = Fh.conv_block(nxt, ws[0], spec=s1)
            return (y.float() * 1.25).sum() + (Fh.conv_block(xr, ws[1], spec=sd).float() * 0.5).sum()

        def decoder():
            if via_crop_cat:
                z = Fh.crop_cat(Fh.conv_block(y0, ws[2], gamma=gam, beta=bet, spec=sc), skip)
            else:
                z = Fh.conv_block(y0, ws[2], gamma=gam, beta=bet, spec=sc, cat_with=skip)
            return (z.float() * torch.linspace(0.5, 1.5, z.shape[-1], device=DEV)).sum()
        # autograd runs the later-created branch's backward first
        total = (stage() + decoder()) if order == 'skip_first' else (decoder() + stage())
        calls = []
        orig = h.sum_maps
        h.sum_maps = lambda maps: (calls.append(len(maps)), orig(maps))[1]
        try:
            total.backward()
        finally:
            h.sum_maps = orig
        torch.cuda.synchronize()
        return x.grad.float(), [w.grad.float() for w in ws], calls

    t = {torch.float32: 2e-6, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}[dtype]
    Fh.set_compute_dtype(dtype)
    try:
        for crop in (False, True):
            g_ref, w_ref, c_ref = run(False, crop)
            g, wg, calls = run(True, crop)
            assert rel_err(g, g_ref) < t
            for a, b in zip(wg, w_ref):
                assert rel_err(a, b) < 1e-5
            assert c_ref == [2]
            assert calls == ([] if order == 'skip_first' else [2]), (order, crop, calls)
    finally:
        Fh.set_compute_dtype(torch.bfloat16)


def test_small_elementwise_ops():
    h = hip()
    g = rnd(3, 3, 10, 12, seed=1).to(DEV)
    mask = (torch.rand(3, 1, 10, 12, generator=torch.Generator().manual_seed(2)) * 3).to(torch.uint8).to(DEV)
    assert torch.equal(h.mask_keep(g, mask), g * (mask == 0))
    s = torch.tensor(0.37, device=DEV)
    assert torch.allclose(h.scale_dev(g, s, 2.0), g * s * 2.0, rtol=1e-6, atol=0)
    assert torch.equal(h.scale_dev(g, None), g)
    v = rnd(2, 4, 7, 9, seed=3).to(DEV)
    v[0, :3, 0, 0] = 0.0                                                     # zero vector: 0 / (0 + 1e-10) = 0
    ref = v[:, :3] / (torch.sqrt((v[:, :3] ** 2).sum(1, keepdim=True)) + 1e-10)
    assert torch.allclose(h.unit_normals(v), ref, rtol=2e-6, atol=1e-7)
    z = torch.ones(5, 7, device=DEV)
    h.zero_(z[1:3])
    assert float(z.sum()) == 21.0


def test_combine_losses_and_flat_fanout():
    from rdfc_gan_amd import functional as Fh
    vals = [0.37, 12.5, 3.25e-3, 0.9]
    ws = [1.0, 100.0, 500.0, 300.0]
    ts = [torch.tensor(v, device=DEV, requires_grad=True) for v in vals]
    total, scaled = Fh.combine_losses(list(zip(ts, ws)), post=0.5)
    ref = torch.tensor(0.0)
    acc = None
    for v, w in zip(vals, ws):
        term = torch.tensor(v) * w
        acc = term if acc is None else acc + term
    assert float(total.detach()) == float(acc * 0.5)                          # sequential f32 adds, in order
    assert torch.equal(scaled.cpu(), torch.tensor([torch.tensor(v) * w for v, w in zip(vals, ws)]))
    assert not scaled.requires_grad
    (total * 3.0).backward()
    for t, w in zip(ts, ws):
        assert float(t.grad) == pytest.approx(1.5 * w, rel=1e-6)
    x = rnd(2, 3, 9, 11, seed=4).to(DEV).requires_grad_(True)
    a, b, c = Fh.fanout_flat(x, 3)
    (a.sum() * 2 + (b * b).sum() + c[:, :1].sum() * 7).backward()
    exp = 2 + 2 * x.detach()
    exp[:, :1] += 7
    assert torch.allclose(x.grad, exp, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('affinity', ['TGASS', 'ASS', 'AS', 'TC'])
@pytest.mark.parametrize('off_scale', [0.6, 3.0], ids=['near', 'far'])
def test_nlspn_fused_against_oracle(affinity, off_scale):
    """csrc/nlspn.hip (offset/affinity kernel + the propagation loop as one autograd node) against the oracle's NLSPNProp
    (nlspn_model.py:68-175 restated on the C restatement of upstream's deformable-conv kernels): values, and the gradients
    of the 24-channel guidance conv output, the initial map, the confidence and aff_scale_const.  'far' offsets leave the
    kernels' LDS windows (global-memory fallback) and the image."""
    from oracle.nlspn import NLSPNProp
    from rdfc_gan_amd.nlspn import _Affinity, _Propagate
    B, H, W, T = 2, 19, 70, 4
    oa = rnd(B, 24, H, W, seed=1)
    oa[:, :16] *= off_scale
    conf = torch.sigmoid(rnd(B, 1, H, W, seed=2))
    feat = rnd(B, 1, H, W, seed=3)
    g_out = rnd(B, 1, H, W, seed=4)
    ref = NLSPNProp(8, 3, T, affinity, 0.5, conf_prop=True)
    # oracle: feed the conv output directly (conv_offset_aff replaced by the identity on a 24-channel "guidance")
    ref.conv_offset_aff = torch.nn.Identity()
    oa_r, conf_r, feat_r = (t.clone().requires_grad_(True) for t in (oa, conf, feat))
    off_r, aff_r = ref.offset_affinity(oa_r, conf_r)
    y = feat_r
    for _ in range(T):
        y = ref.mdconv(y, off_r, aff_r, ref.w, ref.b, 1, 1, 1, 1, 1, 64)
    y.backward(g_out)
    # HIP
    oa_pm = oa.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    conf_d, feat_d = conf.to(DEV).requires_grad_(True), feat.to(DEV).requires_grad_(True)
    scale = ref.aff_scale_const.detach().clone().to(DEV).requires_grad_(True)
    w9, b1, wc = torch.ones(1, 1, 3, 3, device=DEV), torch.zeros(1, device=DEV), torch.ones(1, 1, 1, 1, device=DEV)
    off_d, aff_d = _Affinity.apply(oa_pm, conf_d, scale, wc, b1, affinity)
    assert rel_err(off_d, off_r.detach()) < 1e-6 and rel_err(aff_d, aff_r.detach()) < 2e-5
    out, feats = _Propagate.apply(feat_d, off_d, aff_d, w9, b1, T)
    assert feats.shape == (T + 1, B, 1, H, W) and not feats.requires_grad
    assert rel_err(out, y.detach()) < 2e-5
    out.backward(g_out.to(DEV))
    assert rel_err(feat_d.grad, feat_r.grad) < 5e-5
    assert rel_err(conf_d.grad, conf_r.grad) < 5e-5
    assert rel_err(oa_pm.grad.permute(0, 3, 1, 2), oa_r.grad) < 5e-5
    if affinity == 'TGASS':
        assert abs(float(scale.grad) - float(ref.aff_scale_const.grad)) < 5e-5 * max(1.0, abs(float(ref.aff_scale_const.grad)))


def test_nlspn_scatter_keeps_nonfinite_and_tiny_cotangents():
    """prop_bwd_feat accumulates in 64-bit fixed point scaled per workgroup; a NaN cotangent must stay a NaN in the result
    and cotangents far below the float range of the fixed-point scale (1e-30) must not be flushed to zero"""
    from rdfc_gan_amd.nlspn import _Propagate
    B, H, W, T = 1, 12, 70, 2
    feat = rnd(B, 1, H, W, seed=1).to(DEV).requires_grad_(True)
    off = (rnd(B, 18, H, W, seed=2) * 0.4).to(DEV)
    aff = torch.softmax(rnd(B, 9, H, W, seed=3), 1).to(DEV)
    w9, b1 = torch.ones(1, 1, 3, 3, device=DEV), torch.zeros(1, device=DEV)
    out, _ = _Propagate.apply(feat, off, aff, w9, b1, T)
    g = rnd(B, 1, H, W, seed=4).to(DEV)
    (gx,) = torch.autograd.grad(out, feat, g, retain_graph=True)
    (gt,) = torch.autograd.grad(out, feat, g * 1e-30, retain_graph=True)
    assert rel_err(gt * 1e30, gx) < 1e-5                                      # linear in the cotangent down to 1e-30
    gn = g.clone()
    gn[0, 0, 5, 33] = float('nan')
    (gb,) = torch.autograd.grad(out, feat, gn)
    assert torch.isnan(gb).any() and not torch.isnan(gb[0, 0, :, :8]).any()


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
@pytest.mark.parametrize('case', [(3, 29, 38, 64, 64, 3, 1, 1), (2, 30, 76, 64, 128, 3, 2, 1), (2, 20, 128, 64, 128, 1, 2, 0),
                                  (2, 16, 18, 64, 128, 4, 2, 1), (2, 12, 14, 24, 40, 3, 1, 1)], ids=lambda c: 'x'.join(map(str, c)))
def test_dgrad_with_addend_and_reflect_adjoint_with_addend(case, dtype):
    """rdfc_conv2d_dgrad_add == input gradient + addend (the LDS-DMA kernels add in their epilogue, other shapes run a sum pass);
    the addend may be a channel slice of a wider map.  Same for the reflection pad's adjoint."""
    h = hip()
    N, H, W, Ci, Co, k, s, p = case
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    _, wb = h.pack_weight(w.to(DEV), dtype)
    gy = to_pm(rnd(N, Co, Ho, Wo, seed=4), dtype)
    add = to_pm(rnd(N, Ci, H, W, seed=5), dtype, ld_extra=8)
    d = h.conv_desc((N, H, W, Ci), Co, k, s, p, ldx=Ci, ldy=Co, dtype=h.dt(gy))
    plain = h.conv_dgrad(d, gy, wb, None, torch.empty(N, H, W, Ci, dtype=dtype, device=DEV))
    fused = h.conv_dgrad(d, gy, wb, None, torch.empty(N, H, W, Ci, dtype=dtype, device=DEV), addend=add)
    ref = plain.float() + add.float()
    assert rel_err(fused.float(), ref) < {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    if dtype != torch.float32:
        # round 6: an addend whose pieces are 16-byte aligned is fetched in whole rows by the epilogue's flush from an f32 staging tile
        # (common.h: rsf_flush); one that is only 8-byte aligned -- a slice starting at channel 4 -- keeps the 8-byte pieces on the way into
        # the 16-bit tile.  Same arithmetic in the same order: the two must agree bit for bit
        odd = torch.zeros(N, H, W, Ci + 8, dtype=dtype, device=DEV)[..., 4:4 + Ci]
        odd.copy_(add)
        assert odd.data_ptr() % 16 == 8
        fused_odd = h.conv_dgrad(d, gy, wb, None, torch.empty(N, H, W, Ci, dtype=dtype, device=DEV), addend=odd)
        assert torch.equal(fused_odd, fused)
    # two maps (rdfc_conv2d_dgrad_add2: both in the transposed stride-2 kernel's epilogue where it takes the launch -- RDFC_T3S2=3 in
    # tests/test_variants_gpu.py sends the small shapes there too --, else the one-addend path + a sum pass); the second one a channel slice
    add2 = to_pm(rnd(N, Ci + 8, H, W, seed=7), dtype)[..., 8:]
    fused2 = h.conv_dgrad(d, gy, wb, None, torch.empty(N, H, W, Ci, dtype=dtype, device=DEV), addend=add, addend2=add2)
    assert rel_err(fused2.float(), ref + add2.float()) < {torch.float32: 1e-6, torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}[dtype]
    only2 = h.conv_dgrad(d, gy, wb, None, torch.empty(N, H, W, Ci, dtype=dtype, device=DEV), addend2=add2)
    assert rel_err(only2.float(), plain.float() + add2.float()) < {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    g = to_pm(rnd(N, Ci, H + 4, W + 4, seed=6), dtype)
    a = h.reflect_pad_adjoint(g, 2)
    b = h.reflect_pad_adjoint(g, 2, addend=add)
    assert rel_err(b.float(), a.float() + add.float()) < {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=DT_IDS)
def test_batched_weight_pack_equals_single_packs(dtype):
    """rdfc_pack_weights_batch (all weights of a network in one launch, after the optimizer step) writes exactly what
    rdfc_pack_weight / rdfc_pack_weight_ohwi write one weight at a time: ragged tiles, both master layouts, scales, and the
    registry that drives it (functional._packed / weights_updated) picks up in-place updates of a flat buffer."""
    h = hip()
    from rdfc_gan_amd import functional as Fh
    shapes = [(1, 3, 7, 7), (64, 64, 3, 3), (70, 130, 3, 3), (512, 1, 4, 4), (3, 200, 1, 1), (129, 65, 2, 2)]
    n = sum(int(np.prod(s)) for s in shapes)
    flat = (torch.arange(n, dtype=torch.float32) * 0.37).sin().to(DEV)
    ws, off = [], 0
    for k, s in enumerate(shapes):
        v = flat[off:off + int(np.prod(s))]
        off += int(np.prod(s))
        if k % 2 == 0:          # channels-last storage, logical OIHW (how FlatAdam keeps conv weights)
            O, I, KH, KW = s
            ws.append(v.view(O, KH, KW, I).permute(0, 3, 1, 2))
        else:
            ws.append(v.view(s))
    Fh.invalidate_packed_weights()
    scales = [1.0, 0.5, 1.0, 2.0, 1.0, 0.25]
    first = [Fh._packed(w, dtype, sc) for w, sc in zip(ws, scales)]
    ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in first]
    flat.data.mul_(-1.5).add_(0.25)                # "the optimizer step": the weights change behind the registry's back (no version bump)
    Fh.weights_updated(flat)
    again = [Fh._packed(w, dtype, sc) for w, sc in zip(ws, scales)]
    assert [(a.data_ptr(), b.data_ptr()) for a, b in again] == ptrs          # same buffers, refreshed by the batch launch
    for w, sc, (wf, wb) in zip(ws, scales, again):
        if h.is_ohwi(w) and not w.is_contiguous():
            rf, rb = h.pack_weight_ohwi(w, dtype, sc)
        else:
            rf, rb = h.pack_weight(w.contiguous(), dtype, sc)
        assert torch.equal(wf, rf) and torch.equal(wb, rb), tuple(w.shape)
        O, I, KH, KW = w.shape
        ref = (w.float() * sc).to(dtype)
        assert torch.equal(wf.view(O, KH, KW, I).permute(0, 3, 1, 2), ref)
        assert torch.equal(wb.view(I, KH, KW, O).permute(3, 0, 1, 2), ref)
    Fh.invalidate_packed_weights()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('case', [(2, 57, 76, 256, 256), (3, 29, 38, 64, 128), (1, 17, 40, 128, 64), (2, 8, 20, 64, 64)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_reflection_pad_folded_into_conv_loaders(case, dtype):
    """rdfc_conv2d_forward_reflect / rdfc_conv2d_wgrad_reflect (ReflectionPad2d(1) folded into the LDS-DMA loaders) against the
    explicit pad pass followed by the pad-0 conv: the forward reads the same values in the same order (bit-equal), the weight
    gradient differs by the order of its split-K float atomics only.  f32 maps are not covered (the caller pads explicitly)."""
    h = hip()
    N, H, W, Ci, Co = case
    x = to_pm(rnd(N, Ci, H, W, seed=1), dtype, ld_extra=8)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5)
    wf, _ = h.pack_weight(w.to(DEV), dtype)
    bias = rnd(Co, seed=3).to(DEV)
    d_r = h.conv_desc((N, H, W, Ci), Co, 3, 1, 1, ldx=h.ld(x), ldy=Co, dtype=h.dt(x))
    assert h.conv_reflect_supported(d_r)
    assert not h.conv_reflect_supported(h.conv_desc((N, H, W, Ci), Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=h.F32))
    xp = h.reflect_pad(x.contiguous(), 1)
    d_p = h.conv_desc(tuple(xp.shape), Co, 3, 1, 0, ldx=Ci, ldy=Co, dtype=h.dt(x))
    y_ref = h.conv_forward(d_p, xp, wf, bias, torch.empty(N, H, W, Co, dtype=dtype, device=DEV))
    y = h.conv_forward(d_r, x, wf, bias, torch.empty(N, H, W, Co, dtype=dtype, device=DEV), reflect=True)
    assert torch.equal(y, y_ref)
    ref = F.conv2d(F.pad(from_pm(x), (1, 1, 1, 1), mode='reflect'), w.to(dtype).float(), bias.cpu())
    assert rel_err(from_pm(y), ref) < tol(dtype)
    gy = to_pm(rnd(N, Co, H, W, seed=4), dtype)
    dw_ref = h.conv_wgrad(d_p, xp, gy)
    dw = h.conv_wgrad(h.conv_desc((N, H, W, Ci), Co, 3, 1, 1, ldx=h.ld(x), ldy=Co, dtype=h.dt(x)), x, gy, reflect=True)
    assert rel_err(dw, dw_ref) < 1e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('case', [(4, 134, 146, 64, 3, 7, 0, 'tanh', True), (2, 190, 190, 32, 4, 7, 3, 'none', False),
                                  (3, 160, 150, 96, 2, 5, 2, 'lrelu', False), (1, 260, 270, 64, 1, 4, 1, 'none', True)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_wide_kernel_few_output_channels_row_gemm(case, dtype):
    """conv_thin_row.hip (ResnetGenerator's 7x7 -> 3 output layer, resnet_generator.py:50-52): the conv as a 1x1 MFMA product per
    kernel row followed by a shift-and-add, against torch's conv2d on identically rounded operands; the launch must have gone
    to that kernel."""
    h = hip()
    N, H, W, Ci, Co, k, pad, act, f32out = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), b, 1, pad)
    ref = {'tanh': torch.tanh, 'none': lambda t: t, 'lrelu': lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    x_pm = to_pm(x, dtype, ld_extra=8)
    wf, _ = h.pack_weight(w.to(DEV), dtype)
    Ho, Wo = ref.shape[2:]
    y = torch.empty(N, Ho, Wo, Co, dtype=torch.float32 if f32out else dtype, device=DEV)
    d = h.conv_desc(x_pm.shape, Co, k, 1, pad, ldx=h.ld(x_pm), ldy=Co, dtype=h.dt(x_pm), act=act, slope=0.2, out_f32=f32out)
    h.conv_forward(d, x_pm, wf, b.to(DEV), y)
    assert h.lib().rdfc_last_kernel().decode().startswith('thin_row_kernel')
    assert rel_err(from_pm(y), ref) < tol(dtype)


def test_reserve_streams_creates_the_step_streams_once():
    """functional.reserve_streams (called before init_process_group so that the compute streams own their hardware queues): the
    streams it touches are the ones the step later uses, and calling it twice is harmless."""
    from rdfc_gan_amd import functional as Fh
    dev = torch.device('cuda', torch.cuda.current_device())
    Fh.reserve_streams(dev, with_net=True)
    side, branch, net = Fh.wgrad_stream(dev), Fh.branch_stream(dev), Fh.net_stream(dev)
    Fh.reserve_streams('cuda')
    assert Fh.wgrad_stream(dev) is side and Fh.branch_stream(dev) is branch and Fh.net_stream(dev) is net
    assert len({side.cuda_stream, branch.cuda_stream, net.cuda_stream, torch.cuda.default_stream(dev).cuda_stream}) == 4
    assert set(s.cuda_stream for s in Fh.helper_streams(dev)) == {side.cuda_stream, branch.cuda_stream, net.cuda_stream}
