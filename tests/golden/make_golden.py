"""Generate the committed golden fixtures from the upstream reference.

Run in the BUILD CONTAINER only (needs /root/reference):
    python tests/golden/make_golden.py [--only NAME]
Outputs tests/golden/*.npz.  Inputs and weights are regenerated deterministically
by tests (detweights.py), so fixtures hold expected outputs / gradients /
checksums only.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness as rh                                   # noqa: E402
from detweights import det_state_dict, det_tensor          # noqa: E402
from cases import normal_features, FeatureListAdapter, step_batch_label, normal_loss_case, NLSPN_CFG, gen_inputs, out_weights, grad_summary, step_batch, STEP_HP, metric_case, label_net_weights, LABEL_CASE   # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f'wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)')


def run_module(net, inputs, tag, out_keys=None, full_grads=()):
    """forward (train + eval), then backward of a fixed linear functional of the outputs."""
    res = {}
    net.eval()
    with torch.no_grad():
        o = net(*inputs)
    o = o if isinstance(o, dict) else {'out': o}
    for k, v in o.items():
        res[f'eval.{k}'] = v
    net.train()
    ins = [t.clone().requires_grad_(t.is_floating_point()) for t in inputs]
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
    names, summ = grad_summary(net)
    res['grad_names'] = np.array(names)
    res['grad_summary'] = summ
    for k in full_grads:
        res[f'grad.{k}'] = dict(net.named_parameters())[k].grad
    bufs = {k: v for k, v in net.state_dict().items() if k.endswith('running_mean') or k.endswith('running_var')}
    res['buf_names'] = np.array(list(bufs))
    res['buf_summary'] = np.array([[float(v.sum()), float(v.norm())] for v in bufs.values()])
    return res


def g_a2b(nlspn, B, H, W, name, full=True):
    RG, _, _, _ = rh.ref_classes()
    net = RG(pretrained_on_imagenet=False, use_nlspn_refine=nlspn, nlspn_configs=NLSPN_CFG)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    rgb, depth, normal = gen_inputs('g_a2b', B, H, W)
    fg = ('rgb_branch_en1.0.bias', 'rgb_conf_dec0.0.weight', 'fuse_layer1.style.linear.bias',
          'rgb_branch_encoder_decoder.en6.1.weight', 'depth_branch_en1_depth.0.weight',
          'id_dec0.0.weight', 'depth_branch_encoder_decoder.de2.1.bias')
    if nlspn:
        fg += ('nlspn_refine_module.prop_layer.conv_offset_aff.weight', 'nlspn_refine_module.prop_layer.aff_scale_const')
    res = run_module(net, (rgb, depth, normal), name, full_grads=fg)
    if not full:   # big case: keep train-mode outputs + summaries only
        res = {k: v for k, v in res.items() if not k.startswith('eval.') or k == 'eval.pred_depth'}
    save(name, **res)


def g_b2a(B, H, W, name):
    _, RB, _, _ = rh.ref_classes()
    net = RB(1, 3, 64, 'IN2d')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    x = gen_inputs('g_b2a', B, H, W)[1]
    save(name, **run_module(net, (x,), name, full_grads=('model.23.bias', 'model.6.weight', 'model.23.weight', 'model.18.weight')))


def disc(cin, B, H, W, name):
    _, _, PD, _ = rh.ref_classes()
    net = PD(cin, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    x = det_tensor(3, f'disc{cin}', (B, cin, H, W))
    save(name, **run_module(net, (x,), name, full_grads=('model.0.conv.bias', 'model.4.conv.weight', 'model.1.bn2d.weight')))


def wadain():
    rh.install()
    from lib.models.generator.rdf_generator.model_utils import AdaptiveInstanceNorm
    res = {}
    for i, (c, h, w) in enumerate([(512, 5, 7), (768, 6, 9), (384, 9, 8), (192, 12, 10)], 1):
        m = AdaptiveInstanceNorm(c, c)
        m.load_state_dict(det_state_dict(m.state_dict(), i))
        x = det_tensor(4, f'wadain_x{i}', (2, c, h, w)).requires_grad_(True)
        s = det_tensor(4, f'wadain_s{i}', (2, c, h, w)).requires_grad_(True)
        y = m(x, s)
        (y * out_weights('wadain', str(i), y.shape)).sum().backward()
        res.update({f'y{i}': y, f'gx{i}': x.grad, f'gs{i}': s.grad,
                    f'gw_corner{i}': m.style.linear.weight_orig.grad[:32, :32].clone(),
                    f'gw_summary{i}': np.array([float(m.style.linear.weight_orig.grad.double().sum()),
                                                float(m.style.linear.weight_orig.grad.double().norm())]),
                    f'gb{i}': m.style.linear.bias.grad})
    save('wadain', **res)


def losses():
    _, _, _, gl = rh.ref_classes()
    res = {}
    p = det_tensor(5, 'p', (2, 1, 26, 36)).requires_grad_(True)
    for mode in ('lsgan',):
        crit = gl.GANLoss(mode)
        for real in (True, False):
            v = crit(p, real)
            g, = torch.autograd.grad(v, p)
            res[f'gan.{mode}.{int(real)}'] = v
            res[f'gan.{mode}.{int(real)}.grad'] = g
    for C in (1, 3):
        a = det_tensor(5, f'a{C}', (2, C, 20, 24)).requires_grad_(True)
        b = det_tensor(5, f'b{C}', (2, C, 20, 24))
        m = (det_tensor(5, 'm', (2, 1, 20, 24), 'rand') > 0.3)
        w = m / (m.sum() + 1e-6)
        for nm, fn in (('l1', gl.L1_loss), ('l2', gl.L2_loss)):
            for wn, ww in (('w', w), ('u', None)):
                v = fn(a, b, weight=ww)
                g, = torch.autograd.grad(v, a)
                res[f'{nm}.{C}.{wn}'] = v
                res[f'{nm}.{C}.{wn}.grad'] = g
    save('losses', **res)


class _Node(dict):
    __getattr__ = dict.__getitem__


def step_trace(nsteps=2, label=False, normal=False):
    """Run upstream RDFCGAN.optimize_parameters itself (NNET / PSPNet replaced by
    constant stand-ins) and record the loss dict + weight / BN-buffer checksums.
    label=True: G_label is upstream's real PSPNet (Dropout2d off, see label_net()), the batch carries label maps and a
    normal mask, and every entry of the loss dict is recorded -> step_trace_label.npz."""
    rh.install()
    import lib.models.rdfc_gan as ref_mod
    from lib.models.rdfc_gan import RDFCGAN
    from lib.models.generator.normal_net.NNET import NNET
    stub_psp, stub_nnet = ref_mod.PSPNet, ref_mod.NNET
    if label:
        ref_mod.PSPNet = rh.real_pspnet()
    if normal:      # G_normal = upstream's Decoder behind the stand-in encoder (-> step_trace_full.npz), AdamW with weight decay
        ref_mod.NNET = rh.ref_nnet_with_stand_in_encoder()
    hp = STEP_HP
    model_cfg = _Node(
        G_A2B=dict(type='RDFGenerator', pretrained_on_imagenet=False),
        G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
        D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
        D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'))
    args = _Node(local_rank=0, model=model_cfg, lr=hp['lr'], beta1=hp['beta1'], beta2=hp['beta2'], weight_decay=0.01 if normal else 0.0,
                 gan_loss_type='lsgan', scheduler='linear', epoch=0, decay_epoch=100, n_epochs=150,
                 lambda_A=hp['lambda_A'], lambda_B=hp['lambda_B'], lambda_L1=hp['lambda_L1'], optimizer='adam')
    m = RDFCGAN(device=torch.device('cpu'), distributed=False, args=args, is_train=True,
                out_height=hp['H'], out_width=hp['W'])
    ref_mod.PSPNet, ref_mod.NNET = stub_psp, stub_nnet
    nets = [('G_A2B', m.G_A2B), ('G_B2A', m.G_B2A), ('disc_A', m.disc_A), ('disc_B', m.disc_B)]
    for key, net in nets:
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
    if label:
        m.G_label.load_state_dict(label_net_weights(m.G_label.state_dict()))
        for mod in m.G_label.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        nets.append(('G_label', m.G_label))
    if normal:
        m.G_normal.load_state_dict(det_state_dict(m.G_normal.state_dict(), 0))
        nets.append(('G_normal', m.G_normal))
    m.train()
    res = {}
    for it in range(nsteps):
        data, normal_A, normal_fA = (step_batch_label if label else step_batch)(it, hp['B'], hp['H'], hp['W'])
        pad = torch.zeros(hp['B'], 1, hp['H'], hp['W'])
        if not normal:
            NNET.queue[:] = [torch.cat([normal_A, pad], 1), torch.cat([normal_fA, pad], 1), torch.cat([normal_fA, pad], 1)]
        random.seed(100 + it)
        np.random.seed(100 + it)
        m.set_input(data)
        ls = m.optimize_parameters()
        aux_terms = () if label else ('label', 'normal', 'floor', 'wall', 'ceiling')
        keys = sorted(k for k in ls if not any(s in k for s in aux_terms))
        res[f'loss_names.{it}'] = np.array(keys)
        res[f'loss_values.{it}'] = np.array([ls[k] for k in keys], dtype=np.float64)
        extra = sum(ls[k] for k in ls if any(s in k for s in aux_terms))
        res[f'loss_extra.{it}'] = np.float64(extra)
        for key, net in nets:
            sd = net.state_dict()
            names = [k for k in sd if not k.endswith('num_batches_tracked')]
            res[f'sd_names.{key}'] = np.array(names)
            res[f'sd_summary.{key}.{it}'] = np.array([[float(sd[k].double().sum()), float(sd[k].double().norm())] for k in names])
        res[f'fake_B.{it}'] = m.fake_B.detach().clone()
        res[f'fake_A.{it}'] = m.fake_A.detach().clone()
    save('step_trace_full' if normal else ('step_trace_label' if label else 'step_trace'), **res)


def metric():
    """RDFGANMetric.evaluate_all (lib/metrics/rdf_gan_metric.py:59-151) on results built the way Eval.inference builds them
    (lib/evaluator/evaluator.py:17-46: x * std + mean with std = mean = 5, squeeze, one dict per image)."""
    rh.install()
    from lib.metrics.rdf_gan_metric import RDFGANMetric
    gt_n, pd_n, ev = metric_case()
    std = mean = 5.0
    real, fake = gt_n * std + mean, pd_n * std + mean
    plain = [dict(gt=real[k].squeeze(), pd=fake[k].squeeze()) for k in range(len(real))]
    masked = [dict(gt=real[k].squeeze(), pd=fake[k].squeeze(), evaluate_mask=ev[k].squeeze()) for k in range(len(real))]
    m = RDFGANMetric()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        a = m.evaluate_all(plain)
        b = m.evaluate_all(masked)
        per_image = [m.evaluate_all([r]) for r in masked]
    names = m.metric_name
    save('metric', names=np.array(names), plain=np.array([a[n] for n in names], dtype=np.float64),
         masked=np.array([b[n] for n in names], dtype=np.float64),
         per_image=np.array([[r[n] for n in names] for r in per_image], dtype=np.float64))


def label_net(B, H, W, name):
    """upstream's PSPNet (G_label) with the deterministic weights: eval logits, train-mode [out, aux] and the gradients of
    a fixed linear functional of both.  Dropout2d is switched off (p = 0) in the reference instance: its mask comes from
    the framework's RNG stream, which no other implementation can reproduce; the arithmetic around it is pinned."""
    PSP = rh.real_pspnet()
    net = PSP(LABEL_CASE['classes'])
    net.load_state_dict(label_net_weights(net.state_dict()))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0

    class AsDict(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.n = n

        def forward(self, x):
            o = self.n(x)
            return dict(out=o[0], aux=o[1]) if isinstance(o, list) else dict(out=o)

    rgb = gen_inputs('label_net', B, H, W)[0]
    fg = ('n.master_branch.1.weight', 'n.master_branch.1.bias', 'n.auxiliary_branch.4.weight', 'n.initial.0.0.weight',
          'n.layer1.0.bn1.weight', 'n.layer3.5.bn3.bias')   # (not bottleneck.0.bias: in front of a BN its gradient is analytically 0)
    save(name, **run_module(AsDict(net), (rgb,), name, full_grads=fg))


def normal_losses():
    """upstream's nor_loss / manhattan_loss (lib/losses/gan_loss.py:122-132, 50-99) with RDFCGAN's label ids (rdfc_gan.py:32-34)"""
    _, _, _, gl = rh.ref_classes()
    pred, gt, mask, logits = normal_loss_case()
    pred.requires_grad_(True)
    res = dict(nor=gl.nor_loss(pred, gt, mask), nor_nomask=gl.nor_loss(pred, gt, torch.ones_like(mask)))
    f, w, c = gl.manhattan_loss(pred, logits, mask, 12, 5, 3, 100.0)
    res.update(floor=f, wall=w, ceiling=c)
    # the way rdfc_gan.py:242-250 combines them (lambda_L1 = 100): gradient with respect to the predicted normals
    (res['nor'] * 200.0 + f + w + c).backward()
    res['grad_pred'] = pred.grad.clone()
    pred = pred.detach()
    res = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()}
    f2, w2, c2 = gl.manhattan_loss(pred, logits, mask, 12, 5, 99, 100.0)     # no pixel of class 99: python 0.0 * weight
    res.update(ceiling_absent=np.float32(c2), classes_present=np.bincount(logits.argmax(1).flatten().numpy(), minlength=14))
    save('normal_losses', **res)


def normal_decoder():
    """upstream's own Decoder (normal_net/submodules/decoder.py:6-25) on seeded feature maps: eval / train outputs, gradients
    with respect to the five feature maps, parameter-gradient norms, BN buffers"""
    rh.install()
    import importlib
    Dec = importlib.import_module('lib.models.generator.normal_net.submodules.decoder').Decoder
    net = Dec()
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    fg = ('n.conv3.weight', 'n.conv3.bias', 'n.up4._net.4.weight', 'n.conv2.bias', 'n.up1._net.1.bias')   # (conv biases in front of a BN: zero gradient)
    save('normal_decoder_small', **run_module(FeatureListAdapter(net), tuple(normal_features()), 'normal_decoder_small', full_grads=fg))


JOBS = {
    'g_a2b_small': lambda: g_a2b(False, 2, 60, 76, 'g_a2b_small'),
    'g_a2b_nlspn_small': lambda: g_a2b(True, 2, 60, 76, 'g_a2b_nlspn_small'),
    'g_a2b_full_b1': lambda: g_a2b(False, 1, 228, 304, 'g_a2b_full_b1', full=False),
    'g_b2a_small': lambda: g_b2a(2, 60, 76, 'g_b2a_small'),
    'd_a_small': lambda: disc(1, 2, 60, 76, 'd_a_small'),
    'd_b_small': lambda: disc(3, 2, 60, 76, 'd_b_small'),
    'wadain': wadain,
    'losses': losses,
    'step_trace': step_trace,
    'metric': metric,
    'normal_losses': normal_losses,
    'normal_decoder_small': normal_decoder,
    'step_trace_label': lambda: step_trace(2, label=True),
    'step_trace_full': lambda: step_trace(2, label=True, normal=True),
    'label_net_small': lambda: label_net(LABEL_CASE['B'], LABEL_CASE['H'], LABEL_CASE['W'], 'label_net_small'),
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    a = ap.parse_args()
    torch.set_num_threads(8)
    for name, job in JOBS.items():
        if a.only and name not in a.only:
            continue
        job()
