"""CPU simulation: what does 16-bit STORAGE cost the eval path's dense-depth RMSE (BASELINE.json: "within 1e-3 of reference")?

Runs the oracle's G_A2B (plain torch.nn, eval mode, 228x304, the weights / inputs of tests/test_modules_gpu.py's end-to-end test)
with the roundings the HIP path applies -- packed conv / EqualLR weights, conv outputs in front of a norm, normalised + activated
maps, W-AdaIN's style product and output; the four output heads stay f32 like `out_f32` -- switched on one group at a time, and
reports the output error and the RMSE deviation against the unrounded run.  No GPU, nothing of rdfc_gan_amd/ is imported.

    python tools/lowp_sim.py            # -> the table committed as profiles/r03_lowp_sim.txt

Findings (round 3): in bf16 the deviation (1.2 - 1.5e-3 m) comes from rounding the WEIGHTS, a fixed perturbation of the function
that shifts the mean prediction by ~3e-4 (normalised) whatever the activations are stored in (weights alone 1.47e-3, activations
alone 9e-5); which layers carry it changes with the weights (heads, en2, the fusion GEMMs, ...: a random draw of that size), so
keeping single layers in f32 does not bound it.  f16 (11 significant bits) stays <= 3.3e-4 over five weight seeds.
"""
import copy
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from cases import gen_inputs
from detweights import det_state_dict
from oracle import metrics as om
from oracle import nets as on


def rd(t, dt, on_=True):
    return t.to(dt).float() if on_ else t


def round_weights(net, dt, pick=lambda name: True):
    """conv weights and the scaled EqualLR weights of the modules `pick` selects, rounded to dt (in place)"""
    with torch.no_grad():
        for name, m in net.named_modules():
            if not pick(name):
                continue
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                m.weight.copy_(rd(m.weight, dt))
            if isinstance(m, on._EqLinear):
                s = math.sqrt(2.0 / m.weight_orig.shape[1])
                m.weight_orig.copy_(rd(m.weight_orig * s, dt) / s)
    return net


def forward(net, inputs, dt, round_y=True, round_z=True, round_in=True):
    """eval forward with activations rounded to dt where the HIP path stores them"""
    names = {m: n for n, m in net.named_modules()}
    heads = {n for m, n in names.items() if isinstance(m, nn.Conv2d) and '_dec0' in n}
    hooks = []
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and names[m] not in heads:
            hooks.append(m.register_forward_hook(lambda m_, i, o: rd(o, dt, round_y)))
        if isinstance(m, nn.LeakyReLU):
            hooks.append(m.register_forward_hook(lambda m_, i, o: rd(o, dt, round_z)))

    def bb_forward(self, x):            # BasicBlock / W-AdaIN use functional ops: restated with the roundings in place
        y = rd(F.relu(self.bn1(self.conv1(x))), dt, round_z)
        y = self.bn2(self.conv2(y))
        sc = x if self.downsample is None else self.downsample(x)
        return rd(F.relu(y + sc), dt, round_z)

    def wa_forward(self, x, style):
        gb = rd(self.style.linear(style.permute(0, 2, 3, 1)).permute(0, 3, 1, 2), dt, round_y)
        return rd(gb[:, :self.c] * on.instance_norm(x) + gb[:, self.c:], dt, round_z)

    ob, ow = on.BasicBlock.forward, on.WAdaIN.forward
    on.BasicBlock.forward, on.WAdaIN.forward = bb_forward, wa_forward
    try:
        with torch.no_grad():
            return net(*(rd(t, dt, round_in) for t in inputs))
    finally:
        on.BasicBlock.forward, on.WAdaIN.forward = ob, ow
        for h in hooks:
            h.remove()


def rmse_m(gt, pred):
    return np.array([om.image_metrics(om.denormalise(gt[k].numpy()), om.denormalise(pred[k].numpy()), acc=np.float64)[0] for k in range(gt.shape[0])])


def case(wseed, tag, H=228, W=304):
    net = on.RDFGenerator()
    net.load_state_dict(det_state_dict(net.state_dict(), wseed))
    net.eval()
    rgb, raw, normal = gen_inputs(tag, 2, H, W)
    gt = (raw * 0.9 + 0.05).clamp(-1, 1)
    gt[raw <= -1.0] = -1.0
    with torch.no_grad():
        ref = net(rgb, raw, normal)['pred_depth']
    return net, (rgb, raw, normal), gt, ref, rmse_m(gt, ref)


def report(tag, out, ref, gt, r0):
    e = out - ref
    r = rmse_m(gt, out)
    print(f'  {tag:44s} max|e| {float(e.abs().max()):.2e}  rms {float(e.pow(2).mean().sqrt()):.2e}  mean {float(e.mean()):+.2e}  '
          f'|dRMSE| {np.abs(r - r0).max():.2e} m', flush=True)


if __name__ == '__main__':
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    bf, hf = torch.bfloat16, torch.float16
    print('== which rounding costs what (weights seed 0, three input seeds; errors of pred_depth in normalised units, RMSE in metres) ==')
    for tag in ('eval_e2e', 'eval_e2e_b', 'eval_e2e_c'):
        net, inputs, gt, ref, r0 = case(0, tag)
        print(f'[{tag}] reference RMSE {r0.round(3)}')
        for name, dt, kw, rw in (('bf16: weights + activations', bf, {}, True), ('bf16: activations only', bf, {}, False),
                                 ('bf16: weights only', bf, dict(round_y=False, round_z=False, round_in=False), True),
                                 ('bf16: no conv-output rounding', bf, dict(round_y=False), True), ('f16: weights + activations', hf, {}, True)):
            n2 = round_weights(copy.deepcopy(net), dt) if rw else copy.deepcopy(net)
            report(name, forward(n2, inputs, dt, **kw)['pred_depth'], ref, gt, r0)
    print('== bf16 weight rounding by layer group (weights seed 0, activations unrounded) ==')
    net, inputs, gt, ref, r0 = case(0, 'eval_e2e')
    groups = [('all', lambda n: True), ('output heads (*_dec0)', lambda n: '_dec0' in n), ('all but the output heads', lambda n: '_dec0' not in n),
              ('*_dec1', lambda n: '_dec1' in n), ('en1 stems', lambda n: 'en1' in n), ('RGB encoder-decoder', lambda n: 'rgb_branch_encoder_decoder' in n),
              ('depth encoder-decoder', lambda n: 'depth_branch_encoder_decoder' in n), ('W-AdaIN style GEMMs', lambda n: 'fuse' in n)]
    groups += [(lv, (lambda n, lv=lv: ('.' + lv) in n)) for lv in ('en2', 'en3', 'en4', 'en5', 'en6', 'de5', 'de4', 'de3', 'de2')]
    for name, pick in groups:
        n2 = round_weights(copy.deepcopy(net), bf, pick)
        with torch.no_grad():
            report(name, n2(*inputs)['pred_depth'], ref, gt, r0)
    print('== five weight seeds: is the deviation a property of the format or of the draw? ==')
    for wseed in range(5):
        net, inputs, gt, ref, r0 = case(wseed, f'eval_e2e{wseed}')
        print(f'[weights seed {wseed}] reference RMSE {r0.round(3)}')
        for name, dt, keep in (('bf16: weights + activations', bf, ()), ('bf16, output-head weights f32', bf, ('_dec0',)),
                               ('bf16, output-head + stem weights f32', bf, ('_dec0', 'en1')), ('f16: weights + activations', hf, ())):
            n2 = round_weights(copy.deepcopy(net), dt, lambda n: not any(k in n for k in keep))
            report(name, forward(n2, inputs, dt)['pred_depth'], ref, gt, r0)
