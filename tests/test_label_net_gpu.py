"""GPU: the HIP-backed G_label (rdfc_gan_amd.label_net.PSPNet) against the fixture generated from upstream's PSPNet
(tests/golden/make_golden.py::label_net) and, at the bench resolution, against the CPU oracle's eval logits."""
import numpy as np
import pytest
import torch

from cases import LABEL_CASE, gen_inputs, label_net_weights
from test_modules_gpu import DEV, DT, IDS, TOLS, check, run, setup_dtype

pytestmark = pytest.mark.gpu


class AsDict(torch.nn.Module):
    """the fixture's view of the net: list output -> dict, parameters under 'n.'"""

    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        o = self.n(x)
        return dict(out=o[0], aux=o[1]) if isinstance(o, list) else dict(out=o)


def build(dtype):
    from rdfc_gan_amd.label_net import PSPNet
    setup_dtype(dtype)
    net = PSPNet(LABEL_CASE['classes'])
    net.load_state_dict(label_net_weights(net.state_dict()))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net.to(DEV)


# logits: std 0.5, |max| 2.8.  Gradients: upstream's own f32 input gradient is 6e-4 (relative L2) away from f64 on this case.


@pytest.mark.parametrize('dtype', DT[:2], ids=IDS[:2])
def test_label_net_against_upstream_fixture(golden, dtype):
    net = build(dtype)
    x = gen_inputs('label_net', LABEL_CASE['B'], LABEL_CASE['H'], LABEL_CASE['W'])[0].to(DEV)
    wrapped = AsDict(net)
    res = run(wrapped, (x,), 'label_net_small')
    assert res['train.out'].shape == (LABEL_CASE['B'], LABEL_CASE['classes'], LABEL_CASE['H'], LABEL_CASE['W'])
    assert res['train.out'].dtype == torch.float32
    print(check(wrapped, res, golden('label_net_small'), **TOLS['label_net'][dtype]))


def test_label_net_state_dict_keys_are_upstreams(golden):
    from rdfc_gan_amd.label_net import PSPNet
    g = golden('label_net_small')
    net = PSPNet(14)
    names = {k for k, _ in net.named_parameters()}
    assert names == {str(n)[2:] for n in g['grad_names']}
    assert {str(n)[2:] for n in g['buf_names']} <= set(net.state_dict())


def test_label_net_argmax_agreement_full_resolution():
    """228x304 (the bench resolution; 29x38 trunk maps, all four pyramid bins distinct): bf16 class decisions against the
    CPU oracle's -- the quantity manhattan_loss consumes (gan_loss.py:52)"""
    from oracle.label_net import PSPNet as Oracle
    net = build(torch.bfloat16).eval()
    ref = Oracle(LABEL_CASE['classes'])
    ref.load_state_dict(label_net_weights(ref.state_dict()))
    ref.eval()
    x = gen_inputs('label_full', 1, 228, 304)[0]
    with torch.no_grad():
        want = ref(x)
        got = net(x.to(DEV)).float().cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 0.1 * float(want.abs().max())
    top2 = want.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 0.05 * float(want.abs().max())          # pixels whose winner is not a near-tie
    agree = (got.argmax(1) == want.argmax(1))[decided].float().mean()
    assert float(agree) > 0.995 and float(decided.float().mean()) > 0.5


def test_train_step_with_label_net_against_upstream_trace(golden):
    """Two steps of RDFCGAN(use_label_net=True), f32 mode, against upstream's own optimize_parameters with its real PSPNet
    (tests/golden/step_trace_label.npz): every entry of the loss dict -- label cross entropies, nor_loss and the three
    manhattan terms included -- the fake images and the weights of all five networks."""
    import random
    from cases import STEP_HP, noise_driven, step_batch_label
    from detweights import det_state_dict
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    from test_modules_gpu import _step_cfg
    setup_dtype(torch.float32)
    g = golden('step_trace_label')
    hp = STEP_HP
    cfg = _step_cfg(hp)
    cfg.update(scheduler='linear', epoch=0, decay_epoch=100, n_epochs=150, lambda_A=hp['lambda_A'], lambda_B=hp['lambda_B'],
               lambda_L1=hp['lambda_L1'])
    normals = {}
    m = RDFCGAN(torch.device(DEV), False, cfg, is_train=True, use_label_net=True, normal_provider=lambda tag, img: normals[tag])
    for key, net in m.models.items():
        net.load_state_dict(label_net_weights(net.state_dict()) if key == 'G_label' else det_state_dict(net.state_dict(), 0))
    for mod in m.G_label.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    Fh.invalidate_packed_weights()
    m.train()
    for it in range(2):
        data, nA, nfA = step_batch_label(it, hp['B'], hp['H'], hp['W'])
        normals.update(real_A=nA.to(DEV), fake_A=nfA.to(DEV), fake_A_again=nfA.to(DEV))
        random.seed(100 + it)
        np.random.seed(100 + it)
        m.set_input(data)
        ls = m.optimize_parameters()
        ref = dict(zip((str(k) for k in g[f'loss_names.{it}']), g[f'loss_values.{it}']))
        assert sorted(ls) == sorted(ref)
        for k, v in ref.items():
            # step 0 is a function of the inputs alone; afterwards Adam's first updates (~ lr * g / |g|) turn rounding-level
            # gradient differences into lr-sized weight differences (same bounds as test_train_step_against_upstream_trace)
            lim = 2e-3 * abs(v) + 1e-4 if it == 0 else 3e-2 * abs(v) + 2e-3
            assert abs(ls[k] - v) <= lim, f'step {it} {k}: {ls[k]} vs {v}'
        for key, net in m.models.items():
            sd = net.state_dict()
            keep = [i for i, k in enumerate(g[f'sd_names.{key}']) if not noise_driven(key, str(k))
                    and not (key == 'G_label' and str(k).endswith('bottleneck.0.bias'))]      # bias in front of a BN: zero gradient
            got = np.array([float(sd[k].double().norm()) for k in g[f'sd_names.{key}']])
            np.testing.assert_allclose(got[keep], g[f'sd_summary.{key}.{it}'][keep, 1], rtol=5e-3 if it == 0 else 1.5e-2, atol=2e-4,
                                       err_msg=f'step {it} {key}')      # second step: Adam's first updates amplify rounding (see above)


def test_label_net_batch_of_one_in_train_mode_raises_like_upstream():
    """the 1x1 pyramid level leaves one value per channel: nn.BatchNorm2d refuses it in training mode (upstream's PSPNet
    cannot train at batch size 1); eval mode works"""
    net = build(torch.bfloat16)
    x = gen_inputs('label_b1', 1, 48, 64)[0].to(DEV)
    net.train()
    with pytest.raises(ValueError, match='Expected more than 1 value per channel'):
        net(x)
    net.eval()
    assert net(x).shape == (1, LABEL_CASE['classes'], 48, 64)


def test_label_net_full_resolution_train_mode_against_oracle():
    """228x304, batch 2, train mode, f32 mode: 57x76 -> 29x38 trunk maps (odd sizes through the stride-2 stem, the max pool and
    layer2), the four pyramid bins on a 29x38 map and the x8 logit upsampling, forward and input gradient against the CPU oracle"""
    from oracle.label_net import PSPNet as Oracle
    net = build(torch.float32).train()
    ref = Oracle(LABEL_CASE['classes'])
    ref.load_state_dict(label_net_weights(ref.state_dict()))
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    ref.train()
    x = gen_inputs('label_full_train', 2, 228, 304)[0]
    w = torch.randn(2, LABEL_CASE['classes'], 228, 304, generator=torch.Generator().manual_seed(9))
    xr = x.clone().requires_grad_(True)
    o_ref, a_ref = ref(xr)
    ((o_ref + 0.4 * a_ref) * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    o, a = net(xg)
    ((o + 0.4 * a) * w.to(DEV)).sum().backward()
    eo, ea = float((o.detach().cpu() - o_ref.detach()).abs().max()), float((a.detach().cpu() - a_ref.detach()).abs().max())
    g, gr = xg.grad.cpu().double(), xr.grad.double()
    eg = float((g - gr).norm() / gr.norm())
    print(f'[parity] out {eo:.2e} aux {ea:.2e} (|max| {float(o_ref.abs().max()):.2f}) grad_in L2 {eg:.2e}')
    assert eo < 5e-4 and ea < 5e-4 and eg < 2e-2          # f32 mode (measured 2.4e-4 / 1.7e-5 / 8e-3); logits reach |max| ~ 4 through 53 train-mode BatchNorms
