"""CPU: the oracle (oracle/) against the golden fixtures produced from the upstream
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from cases import NLSPN_CFG, STEP_HP, noise_driven, gen_inputs, grad_summary, out_weights, pepper_mask, step_batch
from detweights import det_state_dict, det_tensor
from oracle import losses as ol
from oracle import nets
from oracle.step import HotPathStep

TOL = dict(rtol=1e-4, atol=1e-5)


def _run(net, inputs, tag):
    res = {}
    net.eval()
    with torch.no_grad():
        o = net(*inputs)
    o = o if isinstance(o, dict) else {'out': o}
    res.update({f'eval.{k}': v for k, v in o.items()})
    net.train()
    ins = [t.clone().requires_grad_(True) for t in inputs]
    o = net(*ins)
    o = o if isinstance(o, dict) else {'out': o}
    loss = 0
    for k, v in o.items():
        res[f'train.{k}'] = v
        loss = loss + (v * out_weights(tag, k, v.shape)).sum()
    loss.backward()
    for i, t in enumerate(ins):
        if t.grad is not None:
            res[f'grad_in.{i}'] = t.grad
    return res


def _check(net, res, g):
    for k in g.files:
        if k.startswith(('eval.', 'train.')):
            np.testing.assert_allclose(res[k].detach().numpy(), g[k], err_msg=k, **TOL)
        elif k.startswith('grad_in.'):      # fp32 summation order differs from upstream's kernels
            np.testing.assert_allclose(res[k].detach().numpy(), g[k], err_msg=k, rtol=2e-3, atol=5e-4)
        elif k.startswith('grad.'):
            np.testing.assert_allclose(dict(net.named_parameters())[k[5:]].grad.numpy(), g[k], err_msg=k, rtol=1e-3,
                                       atol=1e-5 * float(np.abs(g[k]).max()) + 1e-6)
    names, summ = grad_summary(net)
    assert sorted(names) == sorted(g['grad_names'])
    ref = dict(zip(g['grad_names'], g['grad_summary'][:, 1]))
    # conv biases in front of an InstanceNorm have an analytically zero gradient: what autograd returns
    # for them is rounding noise (1e-5 against 1e+3 for the weights) -> only require "still negligible"
    floor = 1e-6 * max(ref.values())
    for n, s_ in zip(names, summ[:, 1]):
        if ref[n] < floor:
            assert s_ < 100 * floor, f'{n}: analytically-zero gradient came out as {s_}'
        else:
            np.testing.assert_allclose(s_, ref[n], rtol=1e-3, err_msg=n)
    sd = net.state_dict()
    bs = np.array([[float(sd[k].sum()), float(sd[k].norm())] for k in g['buf_names']])
    np.testing.assert_allclose(bs, g['buf_summary'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('nlspn,name', [(False, 'g_a2b_small'), (True, 'g_a2b_nlspn_small')])
def test_g_a2b(golden, nlspn, name):
    net = nets.RDFGenerator(use_nlspn_refine=nlspn, nlspn_configs=NLSPN_CFG)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    rgb, depth, normal = gen_inputs('g_a2b', 2, 60, 76)
    res = _run(net, (rgb, depth, normal), name)
    res.pop('grad_in.0', None)          # rgb is never read by the generator
    _check(net, res, golden(name))


def test_g_b2a(golden):
    net = nets.ResnetGenerator(1, 3, 64, 'IN2d')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    x = gen_inputs('g_b2a', 2, 60, 76)[1]
    _check(net, _run(net, (x,), 'g_b2a_small'), golden('g_b2a_small'))


@pytest.mark.parametrize('cin,name', [(1, 'd_a_small'), (3, 'd_b_small')])
def test_patchgan(golden, cin, name):
    net = nets.PatchGANDiscriminator(cin, activation='LeakyReLU')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    x = det_tensor(3, f'disc{cin}', (2, cin, 60, 76))
    _check(net, _run(net, (x,), name), golden(name))


def test_wadain(golden):
    g = golden('wadain')
    for i, (c, h, w) in enumerate([(512, 5, 7), (768, 6, 9), (384, 9, 8), (192, 12, 10)], 1):
        m = nets.WAdaIN(c, c)
        m.load_state_dict(det_state_dict(m.state_dict(), i))
        x = det_tensor(4, f'wadain_x{i}', (2, c, h, w)).requires_grad_(True)
        s = det_tensor(4, f'wadain_s{i}', (2, c, h, w)).requires_grad_(True)
        y = m(x, s)
        (y * out_weights('wadain', str(i), y.shape)).sum().backward()
        gw = m.style.linear.weight_orig.grad
        for got, key in ((y, f'y{i}'), (x.grad, f'gx{i}'), (s.grad, f'gs{i}'), (gw[:32, :32], f'gw_corner{i}'),
                         (m.style.linear.bias.grad, f'gb{i}')):
            np.testing.assert_allclose(got.detach().numpy(), g[key], err_msg=key, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(float(gw.double().norm()), g[f'gw_summary{i}'][1], rtol=1e-4)


def test_losses(golden):
    g = golden('losses')
    p = det_tensor(5, 'p', (2, 1, 26, 36)).requires_grad_(True)
    crit = ol.GANLoss('lsgan')
    for real in (True, False):
        v = crit(p, real)
        gr, = torch.autograd.grad(v, p)
        np.testing.assert_allclose(float(v), g[f'gan.lsgan.{int(real)}'], rtol=1e-6)
        np.testing.assert_allclose(gr.numpy(), g[f'gan.lsgan.{int(real)}.grad'], rtol=1e-5, atol=1e-9)
    for C in (1, 3):
        a = det_tensor(5, f'a{C}', (2, C, 20, 24)).requires_grad_(True)
        b = det_tensor(5, f'b{C}', (2, C, 20, 24))
        m = det_tensor(5, 'm', (2, 1, 20, 24), 'rand') > 0.3
        w = m / (m.sum() + 1e-6)
        for nm, fn in (('l1', ol.l1_loss), ('l2', ol.l2_loss)):
            for wn, ww in (('w', w), ('u', None)):
                v = fn(a, b, ww)
                gr, = torch.autograd.grad(v, a)
                np.testing.assert_allclose(float(v), g[f'{nm}.{C}.{wn}'], rtol=1e-6)
                np.testing.assert_allclose(gr.numpy(), g[f'{nm}.{C}.{wn}.grad'], rtol=1e-5, atol=1e-9)


def build_step_nets(mod):
    G = mod.RDFGenerator()
    Gb = mod.ResnetGenerator(1, 3, 64, 'IN2d')
    DA = mod.PatchGANDiscriminator(1, activation='LeakyReLU')
    DB = mod.PatchGANDiscriminator(3, activation='LeakyReLU')
    for n in (G, Gb, DA, DB):
        n.load_state_dict(det_state_dict(n.state_dict(), 0))
        n.train()
    return G, Gb, DA, DB


def make_pepper(it):
    def fn(img):
        m = pepper_mask(100 + it, img.shape)
        if m is not None:
            m = torch.from_numpy(np.repeat(m, img.shape[1], axis=1)).to(img.device)
            img[m == 1] = 1
            img[m == 2] = -1
        return img
    return fn


def test_step_trace(golden):
    """The oracle step against a trace of upstream's own RDFCGAN.optimize_parameters."""
    g = golden('step_trace')
    hp = STEP_HP
    G, Gb, DA, DB = build_step_nets(nets)
    st = HotPathStep(G, Gb, DA, DB, lr=hp['lr'], betas=(hp['beta1'], hp['beta2']),
                     lambda_A=hp['lambda_A'], lambda_B=hp['lambda_B'], lambda_L1=hp['lambda_L1'])
    for it in range(2):
        data, nA, nfA = step_batch(it, hp['B'], hp['H'], hp['W'])
        st.set_input(data)
        # upstream re-normalises the (already unit) normals: helper.py:404-408
        def renorm(n):
            return n / (torch.sqrt((n ** 2.0).sum(1, keepdim=True)) + 1e-10)
        normals = {'real_A': renorm(nA), 'fake_A': renorm(nfA), 'fake_A_again': renorm(nfA)}
        ls = st.step(lambda tag: normals[tag], make_pepper(it))
        names = list(g[f'loss_names.{it}'])
        vals = g[f'loss_values.{it}']
        extra = float(g[f'loss_extra.{it}'])
        for k, v in zip(names, vals):
            ref = v - extra if k == 'loss_G' else v          # upstream's loss_G also sums the out-of-scope terms
            # step 0 is a pure function of the inputs; from step 1 on Adam (update ~ lr * g/|g| at its first
            # steps) turns rounding-level gradient differences into lr-sized weight differences
            np.testing.assert_allclose(ls[k], ref, rtol=2e-4 if it == 0 else 2e-2, atol=1e-5 if it == 0 else 1e-3,
                                       err_msg=f'step {it} {k}')
        tol = dict(rtol=1e-3, atol=1e-4) if it == 0 else dict(rtol=0, atol=2e-2)
        np.testing.assert_allclose(st.fake_B.numpy(), g[f'fake_B.{it}'], **tol)
        np.testing.assert_allclose(st.fake_A.numpy(), g[f'fake_A.{it}'], **tol)
        for key, net in (('G_A2B', G), ('G_B2A', Gb), ('disc_A', DA), ('disc_B', DB)):
            sd = net.state_dict()
            keep = [i for i, k in enumerate(g[f'sd_names.{key}']) if not noise_driven(key, str(k))]
            got = np.array([float(sd[k].double().norm()) for k in g[f'sd_names.{key}']])
            np.testing.assert_allclose(got[keep], g[f'sd_summary.{key}.{it}'][keep, 1], rtol=2e-3, atol=1e-4,
                                       err_msg=f'step {it} {key}')


def test_metric_matches_upstream(golden):
    """oracle.metrics (rdf_gan_metric.py:59-151 + evaluator.py:28-29) against the reference's RDFGANMetric.evaluate_all"""
    from oracle import metrics as om
    from cases import metric_case
    g = golden('metric')
    gt_n, pd_n, ev = metric_case()
    real, fake = om.denormalise(gt_n.numpy()), om.denormalise(pd_n.numpy())
    plain, _ = om.evaluate_all([dict(gt=real[k], pd=fake[k]) for k in range(len(real))])
    masked, per = om.evaluate_all([dict(gt=real[k], pd=fake[k], evaluate_mask=ev[k].numpy()) for k in range(len(real))])
    assert [str(n) for n in g['names']] == list(om.NAMES)
    np.testing.assert_allclose([plain[n] for n in om.NAMES], g['plain'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose([masked[n] for n in om.NAMES], g['masked'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(per, g['per_image'], rtol=2e-6, atol=1e-7)


def test_label_net_matches_upstream(golden):
    """oracle.label_net.PSPNet against upstream's own PSPNet (labels_model.py:43-94 over backbone_resnet50.py): same
    state_dict keys, eval logits, train-mode [out, aux], input gradient, per-parameter gradient norms, BN buffers"""
    from oracle.label_net import PSPNet
    from cases import LABEL_CASE, gen_inputs, label_net_weights, out_weights
    g = golden('label_net_small')
    net = PSPNet(LABEL_CASE['classes'])
    net.load_state_dict(label_net_weights(net.state_dict()))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0                                     # as in the fixture (make_golden.label_net)
    x = gen_inputs('label_net', LABEL_CASE['B'], LABEL_CASE['H'], LABEL_CASE['W'])[0]
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(x).numpy(), g['eval.out'], rtol=0, atol=1e-6)
    net.train()
    xr = x.clone().requires_grad_(True)
    out, aux = net(xr)
    np.testing.assert_allclose(out.detach().numpy(), g['train.out'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(aux.detach().numpy(), g['train.aux'], rtol=0, atol=2e-5)
    ((out * out_weights('label_net_small', 'out', out.shape)).sum() + (aux * out_weights('label_net_small', 'aux', aux.shape)).sum()).backward()
    scale = np.abs(g['grad_in.0']).max()
    np.testing.assert_allclose(xr.grad.numpy(), g['grad_in.0'], rtol=0, atol=2e-3 * scale)
    ref = dict(zip((str(n)[2:] for n in g['grad_names']), g['grad_summary'][:, 1]))      # fixture names carry the wrapper's 'n.'
    got = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}
    assert sorted(got) == sorted(ref)
    floor = 1e-6 * max(ref.values())
    for k, v in ref.items():
        if v > floor:
            assert abs(got[k] - v) <= 5e-3 * v, k
    sd = net.state_dict()
    for k, row in zip(g['buf_names'], g['buf_summary']):
        assert abs(float(sd[str(k)[2:]].norm()) - row[1]) <= 1e-5 * abs(row[1]) + 1e-6, k


def test_normal_losses_match_upstream(golden):
    """oracle.losses.nor_loss / manhattan_loss against upstream's (lib/losses/gan_loss.py:122-132, 50-99)"""
    from oracle import losses as ol
    from cases import normal_loss_case
    g = golden('normal_losses')
    pred, gt, mask, logits = normal_loss_case()
    assert abs(float(ol.nor_loss(pred, gt, mask)) - float(g['nor'])) < 1e-6
    assert abs(float(ol.nor_loss(pred, gt, torch.ones_like(mask))) - float(g['nor_nomask'])) < 1e-6
    f, w, c = ol.manhattan_loss(pred, logits, 12, 5, 3, 100.0)
    np.testing.assert_allclose([float(f), float(w), float(c)], [g['floor'], g['wall'], g['ceiling']], rtol=2e-6)
    assert float(ol.manhattan_loss(pred, logits, 12, 5, 99, 100.0)[2]) == float(g['ceiling_absent']) == 0.0


def test_normal_decoder_matches_upstream(golden):
    """oracle.normal_net.Decoder against upstream's Decoder (normal_net/submodules/decoder.py:6-25)"""
    from oracle.normal_net import Decoder
    from cases import FeatureListAdapter, normal_features, out_weights
    from detweights import det_state_dict
    g = golden('normal_decoder_small')
    net = Decoder()
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    w = FeatureListAdapter(net)
    feats = normal_features()
    w.eval()
    with torch.no_grad():
        np.testing.assert_allclose(w(*feats).numpy(), g['eval.out'], rtol=0, atol=1e-5)
    w.train()
    ins = [t.clone().requires_grad_(True) for t in feats]
    out = w(*ins)
    np.testing.assert_allclose(out.detach().numpy(), g['train.out'], rtol=0, atol=1e-5)
    (out * out_weights('normal_decoder_small', 'out', out.shape)).sum().backward()
    for i, t in enumerate(ins):
        ref = g[f'grad_in.{i}']
        np.testing.assert_allclose(t.grad.numpy(), ref, rtol=0, atol=2e-4 * np.abs(ref).max())


def test_normal_losses_gradient_matches_upstream(golden):
    from oracle import losses as ol
    from cases import normal_loss_case
    g = golden('normal_losses')
    pred, gt, mask, logits = normal_loss_case()
    pred.requires_grad_(True)
    f, w, c = ol.manhattan_loss(pred, logits, 12, 5, 3, 100.0)
    (ol.nor_loss(pred, gt, mask) * 200.0 + f + w + c).backward()
    np.testing.assert_allclose(pred.grad.numpy(), g['grad_pred'], rtol=0, atol=2e-6 * np.abs(g['grad_pred']).max())
