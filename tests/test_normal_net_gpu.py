"""GPU: G_normal's decoder on the HIP kernels (rdfc_gan_amd.normal_net) against the fixture generated from upstream's Decoder."""
import pytest
import torch

from cases import FeatureListAdapter, normal_features
from detweights import det_state_dict
from test_modules_gpu import DEV, DT, IDS, check, run, setup_dtype

pytestmark = pytest.mark.gpu

TOLS = {torch.float32: dict(out_atol=2e-4, grad_l2=1e-2, buf_rtol=1e-4),
        torch.bfloat16: dict(out_atol=0.15, grad_l2=0.8, buf_rtol=3e-2)}


@pytest.mark.parametrize('dtype', DT, ids=IDS)
def test_normal_decoder_against_upstream_fixture(golden, dtype):
    from rdfc_gan_amd.normal_net import Decoder
    setup_dtype(dtype)
    net = Decoder()
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    wrapped = FeatureListAdapter(net.to(DEV))
    res = run(wrapped, tuple(t.to(DEV) for t in normal_features()), 'normal_decoder_small')
    assert res['train.out'].shape == (2, 4, 32, 40) and res['train.out'].dtype == torch.float32
    print(check(wrapped, res, golden('normal_decoder_small'), **TOLS[dtype]))


def test_nnet_wrapper_resizes_and_requires_an_encoder():
    """NNET.forward = decoder(encoder(img)) resized to (out_height, out_width) (NNET.py:21-25); upstream's encoder is a
    torch.hub download, here a constructor argument"""
    from cases import NORMAL_FEATS
    from rdfc_gan_amd.normal_net import NNET
    setup_dtype(torch.bfloat16)
    with pytest.raises(ValueError, match='encoder'):
        NNET(48, 64)

    class Enc(torch.nn.Module):                       # stand-in with the real feature shapes, differentiable in its input
        def forward(self, img):
            feats = [None] * 12
            for i, (c, h, w) in NORMAL_FEATS.items():
                base = torch.nn.functional.adaptive_avg_pool2d(img, (h, w)).mean(1, keepdim=True)
                feats[i] = base.expand(-1, c, -1, -1) * torch.linspace(-1, 1, c, device=img.device).view(1, c, 1, 1)
            return feats

    net = NNET(48, 64, encoder=Enc()).to(DEV)
    img = torch.randn(2, 3, 96, 128, device=DEV, requires_grad=True)
    out = net(img)
    assert out.shape == (2, 4, 48, 64) and out.dtype == torch.float32
    out[:, :3].square().sum().backward()
    assert img.grad is not None and torch.isfinite(img.grad).all() and float(img.grad.abs().max()) > 0
