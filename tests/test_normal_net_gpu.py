"""GPU: G_normal's decoder on the HIP kernels (rdfc_gan_amd.normal_net) against the fixture generated from upstream's Decoder."""
import pytest
import torch

from cases import FeatureListAdapter, normal_features
from detweights import det_state_dict
from test_modules_gpu import DEV, DT, IDS, TOLS, check, run, setup_dtype

pytestmark = pytest.mark.gpu



@pytest.mark.parametrize('dtype', DT[:2], ids=IDS[:2])
def test_normal_decoder_against_upstream_fixture(golden, dtype):
    from rdfc_gan_amd.normal_net import Decoder
    setup_dtype(dtype)
    net = Decoder()
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    wrapped = FeatureListAdapter(net.to(DEV))
    res = run(wrapped, tuple(t.to(DEV) for t in normal_features()), 'normal_decoder_small')
    assert res['train.out'].shape == (2, 4, 32, 40) and res['train.out'].dtype == torch.float32
    print(check(wrapped, res, golden('normal_decoder_small'), **TOLS['normal_net'][dtype]))


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


def test_full_objective_step_against_upstream_trace(golden):
    """Two steps of RDFCGAN(use_label_net=True, normal_net=NNET(...)) in f32 mode against upstream's own optimize_parameters
    run with its real PSPNet and its real normal Decoder behind the same stand-in encoder (tests/golden/step_trace_full.npz):
    all 23 loss entries, the weights of all six networks (AdamW with weight decay on G_normal)."""
    import random
    import numpy as np
    from cases import STEP_HP, StandInEncoder, label_net_weights, noise_driven, step_batch_label
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.normal_net import NNET
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    from test_modules_gpu import _step_cfg
    setup_dtype(torch.float32)
    g = golden('step_trace_full')
    hp = STEP_HP
    cfg = _step_cfg(hp)
    cfg.update(scheduler='linear', epoch=0, decay_epoch=100, n_epochs=150, lambda_A=hp['lambda_A'], lambda_B=hp['lambda_B'],
               lambda_L1=hp['lambda_L1'], weight_decay=0.01)
    m = RDFCGAN(torch.device(DEV), False, cfg, is_train=True, use_label_net=True, out_height=hp['H'], out_width=hp['W'],
                normal_net=NNET(hp['H'], hp['W'], encoder=StandInEncoder()))
    for key, net in m.models.items():
        net.load_state_dict(label_net_weights(net.state_dict()) if key == 'G_label' else det_state_dict(net.state_dict(), 0))
    for mod in m.G_label.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    Fh.invalidate_packed_weights()
    m.train()
    for it in range(2):
        data, _, _ = step_batch_label(it, hp['B'], hp['H'], hp['W'])
        random.seed(100 + it)
        np.random.seed(100 + it)
        m.set_input(data)
        ls = m.optimize_parameters()
        ref = dict(zip((str(k) for k in g[f'loss_names.{it}']), g[f'loss_values.{it}']))
        assert sorted(ls) == sorted(ref)
        for k, v in ref.items():
            lim = 2e-3 * abs(v) + 1e-4 if it == 0 else 3e-2 * abs(v) + 2e-3
            if it == 1 and any(t in k for t in ('floor', 'wall', 'ceiling')):
                # sums over arg-max class masks of freshly initialised logits (near-ties everywhere) one Adam step later: a
                # discontinuous function of lr-sized weight differences -- a handful of flipped pixels moves them by percents
                lim = 0.15 * abs(v)
            assert abs(ls[k] - v) <= lim, f'step {it} {k}: {ls[k]} vs {v}'
        for key, net in m.models.items():
            sd = net.state_dict()
            names = [str(k) for k in g[f'sd_names.{key}']]
            keep = [i for i, k in enumerate(names) if not noise_driven(key, k)
                    and not (key == 'G_label' and k.endswith('bottleneck.0.bias'))
                    and not (key == 'G_normal' and k.endswith(('_net.0.bias', '_net.3.bias')))]       # conv biases in front of a BN
            got = np.array([float(sd[k].double().norm()) for k in names])
            np.testing.assert_allclose(got[keep], g[f'sd_summary.{key}.{it}'][keep, 1], rtol=5e-3 if it == 0 else 1.5e-2, atol=2e-4,
                                       err_msg=f'step {it} {key}')
