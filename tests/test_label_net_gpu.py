"""GPU: the HIP-backed G_label (rdfc_gan_amd.label_net.PSPNet) against the fixture generated from upstream's PSPNet
(tests/golden/make_golden.py::label_net) and, at the bench resolution, against the CPU oracle's eval logits."""
import numpy as np
import pytest
import torch

from cases import LABEL_CASE, gen_inputs, label_net_weights
from test_modules_gpu import DEV, DT, IDS, check, run, setup_dtype

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
TOLS = {torch.float32: dict(out_atol=2e-4, grad_l2=1e-2, buf_rtol=1e-4),
        torch.bfloat16: dict(out_atol=0.25, grad_l2=0.8, buf_rtol=3e-2)}


@pytest.mark.parametrize('dtype', DT, ids=IDS)
def test_label_net_against_upstream_fixture(golden, dtype):
    net = build(dtype)
    x = gen_inputs('label_net', LABEL_CASE['B'], LABEL_CASE['H'], LABEL_CASE['W'])[0].to(DEV)
    wrapped = AsDict(net)
    res = run(wrapped, (x,), 'label_net_small')
    assert res['train.out'].shape == (LABEL_CASE['B'], LABEL_CASE['classes'], LABEL_CASE['H'], LABEL_CASE['W'])
    assert res['train.out'].dtype == torch.float32
    print(check(wrapped, res, golden('label_net_small'), **TOLS[dtype]))


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
