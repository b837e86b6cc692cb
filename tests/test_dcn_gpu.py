"""GPU: the non-fused deformable-conv deliverable through the interface upstream binds.

`rdfc_gan_amd.DCN` stands in for upstream's pybind module `DCN` (deformconv/src/modulated_deform_conv.h:10-87); it is driven here
the way upstream drives it -- through `ModulatedDeformConvFunction` with the call pattern of
nlspn/modulated_deform_conv_func.py:26-53 -- and checked against the C restatement of upstream's CUDA kernels (oracle.mdconv),
in float32 and float64 (AT_DISPATCH_FLOATING_TYPES, modulated_deform_conv_cuda.cu:94), with the argument checks of
modulated_deform_conv_cuda.cu:39-62.  Then the step-by-step NLSPN that runs on it (prop_kernel != 3, preserve_input, or
RDFC_NLSPN_STEPWISE=1) against the fused kernels and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got.detach() - ref.detach()).abs().max() / (ref.detach().abs().max() + 1e-12))


# B, C, H, W, Co, k, stride, pad, dil, groups, deformable groups
CASES = [(2, 4, 9, 11, 6, 3, 1, 1, 1, 2, 2),         # deformconv/test.py's shape family: grouped, two offset groups
         (2, 1, 19, 70, 1, 3, 1, 1, 1, 1, 1),        # NLSPN's propagation step (one channel: the LDS-window backward in f32)
         (2, 1, 19, 70, 1, 1, 1, 0, 1, 1, 1),        # NLSPN's confidence gather (1x1, padding 0)
         (1, 6, 10, 12, 4, 3, 2, 1, 2, 1, 3)]        # stride 2, dilation 2, three offset groups


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_function_against_oracle(case, dtype):
    """forward values and all five gradients of ModulatedDeformConvFunction (upstream's argument order) against oracle.mdconv"""
    from oracle.mdconv import MDConvOracle
    from rdfc_gan_amd.nlspn import ModulatedDeformConvFunction
    B, C, H, W, Co, k, s, p, d, g, dg = case
    Ho, Wo = (H + 2 * p - (d * (k - 1) + 1)) // s + 1, (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    inp = rnd(B, C, H, W, seed=1, dtype=dtype)
    off = rnd(B, dg * 2 * k * k, Ho, Wo, seed=2, scale=2.0, dtype=dtype)          # test.py:405-434 draws 2 * N(0, 1)
    msk = torch.sigmoid(rnd(B, dg * k * k, Ho, Wo, seed=3, dtype=dtype))
    wt = rnd(Co, C // g, k, k, seed=4, scale=0.3, dtype=dtype)
    bias = rnd(Co, seed=5, dtype=dtype)
    gout = rnd(B, Co, Ho, Wo, seed=6, dtype=dtype)
    ref_in = [t.clone().requires_grad_(True) for t in (inp, off, msk, wt, bias)]
    ref = MDConvOracle.apply(*ref_in, s, p, d, g, dg, 64)
    ref.backward(gout)
    got_in = [t.clone().to(DEV).requires_grad_(True) for t in (inp, off, msk, wt, bias)]
    got = ModulatedDeformConvFunction.apply(*got_in, s, p, d, g, dg, 64)
    assert got.dtype == dtype and got.shape == ref.shape
    got.backward(gout.to(DEV))
    lim = 2e-5 if dtype == torch.float32 else 1e-12
    assert rel_err(got, ref.detach()) < lim
    for name, a, b in zip(('input', 'offset', 'mask', 'weight', 'bias'), got_in, ref_in):
        assert a.grad is not None and rel_err(a.grad, b.grad) < (4e-5 if dtype == torch.float32 else 1e-11), name


def test_module_entry_points_and_im2col_step_invariance():
    """the pybind-shaped entry points directly (argument order of modulated_deform_conv.h:10-87); the result does not depend on
    im2col_step (deformconv/test.py:219-260, 304-349: the column buffer is never materialised here)"""
    from rdfc_gan_amd import DCN
    B, C, H, W, Co, k = 4, 2, 8, 9, 3, 3
    inp, wt, bias = rnd(B, C, H, W, seed=1).to(DEV), rnd(Co, C, k, k, seed=2).to(DEV), rnd(Co, seed=3).to(DEV)
    off, msk = rnd(B, 2 * k * k, H, W, seed=4).to(DEV), torch.sigmoid(rnd(B, k * k, H, W, seed=5)).to(DEV)
    gout = rnd(B, Co, H, W, seed=6).to(DEV)
    outs, grads = [], []
    for step in (1, 2, 4, 64):
        outs.append(DCN.modulated_deform_conv_forward(inp, wt, bias, off, msk, k, k, 1, 1, 1, 1, 1, 1, 1, 1, step))
        grads.append(DCN.modulated_deform_conv_backward(inp, wt, bias, off, msk, gout, k, k, 1, 1, 1, 1, 1, 1, 1, 1, step))
    assert len(grads[0]) == 5
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    for gs in grads[1:]:
        for a, b in zip(gs[1:], grads[0][1:]):          # offset / mask / weight / bias gradients are gathers: bit-identical
            assert torch.equal(a, b)
        assert rel_err(gs[0], grads[0][0]) < 1e-5      # the input gradient is a scatter of float atomics


def test_argument_checks_follow_upstream():
    """modulated_deform_conv_cuda.cu:39-62 (forward) / :140-175 (backward): RuntimeError, upstream's wording"""
    from rdfc_gan_amd import DCN
    from rdfc_gan_amd.nlspn import ModulatedDeformConvFunction
    B, C, H, W, Co, k = 3, 2, 8, 9, 4, 3
    inp, wt, bias = rnd(B, C, H, W, seed=1).to(DEV), rnd(Co, C, k, k, seed=2).to(DEV), rnd(Co, seed=3).to(DEV)
    off, msk = rnd(B, 2 * k * k, H, W, seed=4).to(DEV), torch.ones(B, k * k, H, W, device=DEV)
    args = (k, k, 1, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match='input tensor has to be contiguous'):
        DCN.modulated_deform_conv_forward(inp.transpose(2, 3).contiguous().transpose(2, 3), wt, bias, off, msk, *args, 64)
    with pytest.raises(RuntimeError, match='input tensor has to be contiguous'):       # upstream's Function passes its input on as it is
        ModulatedDeformConvFunction.apply(inp.transpose(2, 3).contiguous().transpose(2, 3), off, msk, wt, bias, 1, 1, 1, 1, 1, 64)
    with pytest.raises(RuntimeError, match='weight tensor has to be contiguous'):
        DCN.modulated_deform_conv_forward(inp, wt.transpose(2, 3).contiguous().transpose(2, 3), bias, off, msk, *args, 64)
    with pytest.raises(RuntimeError, match=r'batch\(3\) must divide im2col_step\(2\)'):
        DCN.modulated_deform_conv_forward(inp, wt, bias, off, msk, *args, 2)
    with pytest.raises(RuntimeError, match='Input shape and kernel shape wont match'):
        DCN.modulated_deform_conv_forward(inp, wt, bias, off, msk, 5, 5, 1, 1, 1, 1, 1, 1, 1, 1, 64)
    with pytest.raises(RuntimeError, match='Input shape and kernel channels wont match'):
        DCN.modulated_deform_conv_forward(inp, wt, bias, off, msk, k, k, 1, 1, 1, 1, 1, 1, 2, 1, 64)
    with pytest.raises(RuntimeError, match=r'must divide group'):
        DCN.modulated_deform_conv_forward(rnd(B, 3, H, W).to(DEV), rnd(Co, 1, k, k).to(DEV), bias, off, msk, k, k, 1, 1, 1, 1, 1, 1, 3, 1, 64)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        DCN.modulated_deform_conv_forward(inp.cpu(), wt.cpu(), bias.cpu(), off.cpu(), msk.cpu(), *args, 64)
    with pytest.raises(RuntimeError, match='not implemented for'):
        DCN.modulated_deform_conv_forward(inp.half(), wt.half(), bias.half(), off.half(), msk.half(), *args, 64)
    gout = rnd(B, Co, H, W, seed=6).to(DEV)
    with pytest.raises(RuntimeError, match=r'batch\(3\) must divide im2col_step\(2\)'):
        DCN.modulated_deform_conv_backward(inp, wt, bias, off, msk, gout, *args, 2)
    with pytest.raises(RuntimeError, match='Input shape and kernel shape wont match'):
        DCN.modulated_deform_conv_backward(inp, wt, bias, off, msk, gout, 5, 5, 1, 1, 1, 1, 1, 1, 1, 1, 64)


def _nlspn_case(affinity, preserve_input, T, B=2, H=19, W=70, k_f=3, seed=0):
    """inputs shared by the oracle and the HIP modules: the 3 * (k_f^2 - 1)-channel conv weights drawn small so that the
    offsets stay within a few pixels"""
    n = k_f * k_f - 1
    guide = rnd(B, n, H, W, seed=seed + 1)
    init = rnd(B, 1, H, W, seed=seed + 2)
    conf = torch.sigmoid(rnd(B, 1, H, W, seed=seed + 3))
    fix = torch.relu(rnd(B, 1, H, W, seed=seed + 4)) if preserve_input else None        # sparse "observed" depth
    w = rnd(3 * n, n, 3, 3, seed=seed + 5, scale=0.15)
    b = rnd(3 * n, seed=seed + 6, scale=0.1)
    return guide, init, conf, fix, w, b, rnd(B, 1, H, W, seed=seed + 7)


def _run_hip(affinity, preserve_input, T, k_f, case, stepwise, monkeypatch):
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.nlspn import NLPSN
    guide, init, conf, fix, w, b, gout = case
    if stepwise:
        monkeypatch.setenv('RDFC_NLSPN_STEPWISE', '1')
    else:
        monkeypatch.delenv('RDFC_NLSPN_STEPWISE', raising=False)
    Fh.set_compute_dtype(torch.float32)
    m = NLPSN(k_f * k_f - 1, 1, 3, k_f, T, affinity, 0.5, True, preserve_input).to(DEV)
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(w)
        m.conv_offset_aff.bias.copy_(b)
    Fh.invalidate_packed_weights()
    assert m._fused() == (not stepwise and not preserve_input and k_f == 3)
    g_pm = guide.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    init_d, conf_d = init.to(DEV).requires_grad_(True), conf.to(DEV).requires_grad_(True)
    y, feats, off, aff, _ = m(init_d, g_pm, conf_d, fix.to(DEV) if fix is not None else None)
    assert len(feats) == T
    y.backward(gout.to(DEV))
    return dict(y=y.detach(), off=off.detach(), aff=aff.detach(), g_init=init_d.grad, g_conf=conf_d.grad,
                g_guide=g_pm.grad.permute(0, 3, 1, 2), g_w=m.conv_offset_aff.weight.grad.clone(),
                g_scale=m.aff_scale_const.grad.clone() if m.aff_scale_const.grad is not None else None)


def _run_oracle(affinity, preserve_input, T, k_f, case):
    from oracle.nlspn import NLSPNProp
    guide, init, conf, fix, w, b, gout = case
    ref = NLSPNProp(k_f * k_f - 1, k_f, T, affinity, 0.5, conf_prop=True, preserve_input=preserve_input)
    with torch.no_grad():
        ref.conv_offset_aff.weight.copy_(w)
        ref.conv_offset_aff.bias.copy_(b)
    g_r, init_r, conf_r = (t.clone().requires_grad_(True) for t in (guide, init, conf))
    y, off, aff = ref(init_r, g_r, conf_r, fix)
    y.backward(gout)
    return dict(y=y.detach(), off=off.detach(), aff=aff.detach(), g_init=init_r.grad, g_conf=conf_r.grad, g_guide=g_r.grad,
                g_w=ref.conv_offset_aff.weight.grad, g_scale=ref.aff_scale_const.grad)


def _compare(a, b, lim, keys=('y', 'off', 'aff', 'g_init', 'g_conf', 'g_guide', 'g_w', 'g_scale')):
    for k in keys:
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, k
            continue
        assert rel_err(a[k], b[k]) < lim, (k, rel_err(a[k], b[k]))


@pytest.mark.parametrize('affinity', ['TGASS', 'AS'])
def test_stepwise_nlspn_equals_fused_and_oracle(affinity, monkeypatch):
    """the module in its two implementations -- upstream's structure step by step on the DCN entry points (RDFC_NLSPN_STEPWISE=1)
    and the fused kernels of csrc/nlspn.hip -- against each other and against oracle.nlspn (nlspn_model.py:68-175)"""
    T, case = 4, _nlspn_case(affinity, False, 4)
    ref = _run_oracle(affinity, False, T, 3, case)
    step = _run_hip(affinity, False, T, 3, case, True, monkeypatch)
    fused = _run_hip(affinity, False, T, 3, case, False, monkeypatch)
    _compare(step, ref, 6e-5)
    _compare(fused, ref, 6e-5)
    _compare(step, fused, 6e-5)


def test_stepwise_nlspn_preserve_input_against_oracle(monkeypatch):
    """preserve_input=True (nlspn_model.py:158-169: observed pixels are re-imposed before every step) only exists step by step"""
    T, case = 3, _nlspn_case('TGASS', True, 3, seed=20)
    _compare(_run_hip('TGASS', True, T, 3, case, False, monkeypatch), _run_oracle('TGASS', True, T, 3, case), 6e-5)


def test_stepwise_nlspn_prop_kernel_5_against_oracle(monkeypatch):
    """prop_kernel = 5 (24 neighbours): outside the fused kernels' 3x3 coverage, runs on the general deformable conv"""
    T, case = 2, _nlspn_case('TGASS', False, 2, B=1, H=14, W=40, k_f=5, seed=40)
    _compare(_run_hip('TGASS', False, T, 5, case, False, monkeypatch), _run_oracle('TGASS', False, T, 5, case), 6e-5)
