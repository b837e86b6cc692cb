"""GPU: the HIP-backed modules (rdfc_gan_amd) against the golden fixtures generated from the
upstream reference, and against the CPU oracle on the same seeded inputs.

f32 mode (exact-f32 MFMA) is the parity mode: tolerances below are fp32-reordering sized.
bf16 mode (the benchmarked configuration) is checked against the same references with
bf16-sized tolerances, including the north-star acceptance metric (dense-depth RMSE
within 1e-3 of the reference).
"""
import os
import numpy as np
import pytest
import torch

from cases import NLSPN_CFG, STEP_HP, noise_driven, gen_inputs, grad_summary, out_weights, step_batch
from detweights import det_state_dict, det_tensor

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def setup_dtype(dtype):
    from rdfc_gan_amd import functional as Fh
    Fh.set_compute_dtype(dtype)
    Fh.invalidate_packed_weights()


def run(net, inputs, tag):
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
        loss = loss + (v * out_weights(tag, k, v.shape).to(DEV)).sum()
    loss.backward()
    for i, t in enumerate(ins):
        if t.grad is not None:
            res[f'grad_in.{i}'] = t.grad
    return res


def err(a, b):
    """(max abs error, relative L2 error)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()), float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def check(net, res, g, out_atol, grad_in_l2, grad_l2, norm_median, norm_max, buf_rtol, stock=None, out_rms=None, grad_scalar=None):
    """collect every metric first (printed with -s / on failure), then assert.

    Outputs: max abs error (maps live in [-1, 1]).  Gradients: relative L2 error -- an
    elementwise bound is meaningless here: wherever a pre-activation sits within rounding of 0,
    ReLU / LeakyReLU masks legitimately differ between two fp32 summation orders (the reference
    itself is 7e-3 max-rel / 1.4e-3 L2-rel away from an fp64 evaluation of the same net).
    Limits per tensor class: outputs | input gradients | sampled full parameter gradients | median and max relative
    error of the parameter-gradient norms | BatchNorm buffers.
    stock: {key: relative L2 error of a stock-PyTorch evaluation of the oracle net in the same storage precision}: the
    input gradients here may not be worse than 1.3x that (+ 0.02).
    grad_scalar: limit for sampled gradients that are ONE number (a PReLU slope: the sum of dz * u over every negative unit of a map, with
    heavy cancellation -- in f32 mode it already sits 5.6e-3 from the oracle, i.e. ~1e5 roundings' worth); default: grad_l2.
    norm_max=None / grad_scalar='seeds': the statistic is printed but not asserted on this one fixture -- in the 16-bit modes the MAXIMUM
    over ~340 gradient norms and a single heavily cancelling scalar move several-fold with the summation order of one kernel (round 5) and,
    in a stock PyTorch evaluation, with the weight seed (0.13 ... 0.32 and 0.07 ... 0.83: tests/golden/make_stock_lowp.py); they are
    asserted over six weight seeds against that stock evaluation by test_noisy_lowp_statistics_over_weight_seeds."""
    report, bad = [], []

    def note(k, e, lim):
        report.append(f'{k}={e:.2e}')
        if not e < lim:
            bad.append(f'{k}: {e:.3e} >= {lim}')
    for k in g.files:
        if k.startswith(('eval.', 'train.')):
            note(k, err(res[k].detach().cpu().numpy(), g[k])[0], out_atol)
            if out_rms is not None:          # (where a max over 9 120 pixels says little: bf16 guidance in front of 18 sampling steps)
                d = res[k].detach().cpu().numpy().astype(np.float64) - g[k]
                note(k + '/rms', float(np.sqrt(np.mean(d * d))), out_rms)
        elif k.startswith('grad_in.'):
            e = err(res[k].detach().cpu().numpy(), g[k])[1]
            note(k, e, grad_in_l2)
            if stock is not None and k in stock:
                note(k + '/stock', e / max(stock[k], 1e-9), 1.3 + 0.02 / max(stock[k], 1e-9))
        elif k.startswith('grad.'):
            got = dict(net.named_parameters())[k[5:]].grad.detach().cpu().numpy()
            if g[k].size == 1 and grad_scalar == 'seeds':
                report.append(f'{k}={err(got, g[k])[1]:.2e} (one number: asserted over weight seeds)')
            else:
                note(k, err(got, g[k])[1], grad_scalar if (grad_scalar is not None and g[k].size == 1) else grad_l2)
    names, summ = grad_summary(net)
    ref = dict(zip(g['grad_names'], g['grad_summary'][:, 1]))
    assert sorted(names) == sorted(ref)
    floor = 1e-6 * max(ref.values())          # below: analytically-zero gradients (bias in front of an IN)
    rels = {}
    for n, s_ in zip(names, summ[:, 1]):
        if ref[n] < floor:
            if not s_ < 1000 * floor:
                bad.append(f'{n}: analytically-zero gradient came out as {s_:.3e}')
        else:
            rels[n] = abs(s_ - ref[n]) / ref[n]
    rel = np.array(list(rels.values()))
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:6]
    note('param_grad_norm.median', float(np.median(rel)), norm_median)
    if norm_max is None:
        report.append(f'param_grad_norm.max={float(rel.max()):.2e} (asserted over weight seeds)')
    else:
        note('param_grad_norm.max', float(rel.max()), norm_max)
    report.append('worst=' + ', '.join(f'{k}:{v:.1e}' for k, v in worst))
    sd = net.state_dict()
    if len(g['buf_names']):
        bs = np.array([float(sd[k].norm()) for k in g['buf_names']])
        note('bn_buffers.maxrel', float(np.max(np.abs(bs - g['buf_summary'][:, 1]) / (np.abs(g['buf_summary'][:, 1]) + 1e-4))), buf_rtol)
    print('\n  '.join(['[parity]'] + report))
    assert not bad, '; '.join(bad)
    return report


# Limits <= 2x the values measured on MI355X (round 2, `pytest -s`; measured value in the comment of each entry).
# f32 pins the arithmetic.  bf16 / f16 store every activation and gradient map with 8 / 11 significant bits: a pre-activation
# within that rounding of 0 flips its ReLU / LeakyReLU mask, and each flipped unit changes the gradient path through it --
# the input gradients of these nets are 0.08 (5 layers) to 0.35 (60 layers) away from f32 in ANY bf16 evaluation, stock
# PyTorch's included (_stock_lowp below measures it: D_A 0.082, G_B2A 0.240, G_A2B 0.346 / 0.339 against 0.076 / 0.22 / 0.31
# here), so besides the absolute limits the bf16 / f16 cases assert "not worse than 1.3x a stock low-precision evaluation".
TOLS = {
    'g_a2b': {torch.float32: dict(out_atol=5e-5, grad_in_l2=8e-3, grad_l2=8e-3, norm_median=2e-4, norm_max=4e-3, buf_rtol=1e-5),
              # f32 measured: out 1.0e-5, grad_in 4.3e-3 (full size) / 1.4e-3, sampled grads 4.7e-3, median 8.1e-5, max 1.9e-3, buffers 1.8e-7
              torch.bfloat16: dict(out_atol=0.15, grad_in_l2=0.6, grad_l2=1.0, norm_median=2e-2, norm_max=None, buf_rtol=2e-3),
              # bf16 measured: out 7.6e-2, grad_in 0.32, sampled grads 0.54 (a 40-sample BatchNorm weight), median 8.8e-3, max 0.106, buffers 8.2e-4
              # (the WORST of the 340 parameter-gradient norms read 0.106 in round 4 and 0.207 in round 5, when the 3x3 kernel's tap order
              #  changed; round 5 widened its limit 0.2 -> 0.25.  Round 6: no single-fixture limit any more -- norm_max=None, see check())
              torch.float16: dict(out_atol=2e-2, grad_in_l2=0.24, grad_l2=0.6, norm_median=5.5e-3, norm_max=None, buf_rtol=3e-4)},
              # f16 measured: out 9.5e-3, grad_in 0.116 (0.91x a stock f16 evaluation), sampled grads 0.29, median 2.3e-3, max 6.8e-2, buffers 1.1e-4
    'g_a2b_nlspn': {torch.float32: dict(out_atol=1e-4, grad_in_l2=2e-3, grad_l2=2e-3, norm_median=2e-4, norm_max=1e-3, buf_rtol=1e-5),
                    # f32 measured with the fused NLSPN kernels: out 4.7e-5 (18 propagation steps), grad_in 6.7e-4, grads 9.5e-4, max 3.0e-4
                    torch.bfloat16: dict(out_atol=0.8, out_rms=3e-2, grad_in_l2=0.6, grad_l2=1.1, norm_median=2e-2, norm_max=1.1, buf_rtol=2e-3),
                    # bf16: offsets come from bf16 guidance features; 18 steps of sampling at slightly different positions move a few pixels
                    # of the train-mode map a long way (max 0.41, 99.9th percentile 0.12, on this fixture's N(0, 0.02) guidance weights the
                    # recursion amplifies rounding: test_shapes_gpu.py::test_config3_*).  The max limit is a ceiling only; what is asserted
                    # for this combination is the RMS error over each output map: measured 1.5e-2 (train.depth_map_2; 8.9e-3 on pred_depth),
                    # limit 2x.  aff_scale_const 0.53
                    torch.float16: dict(out_atol=7e-2, out_rms=4e-3, grad_in_l2=0.2, grad_l2=0.7, norm_median=6e-3, norm_max=8e-2, buf_rtol=3e-4)},
                    # f16 measured: out 3.2e-2 (rms 1.7e-3), grad_in 9.5e-2, sampled grads 0.33, median 2.7e-3, max 3.7e-2, buffers 1.1e-4
    'g_b2a': {torch.float32: dict(out_atol=3e-5, grad_in_l2=7e-3, grad_l2=7e-3, norm_median=4e-5, norm_max=1.1e-2, buf_rtol=1e-5),
              # f32 measured: out 1.25e-5, grad_in 3.5e-3, grads 3.1e-3, median 1.8e-5, max 5.3e-3 (a weight in front of an InstanceNorm)
              torch.bfloat16: dict(out_atol=0.12, grad_in_l2=0.45, grad_l2=5e-2, grad_scalar='seeds', norm_median=3e-3, norm_max=None, buf_rtol=2e-3),
              # bf16 measured: out 5.6e-2, grad_in 0.22, grads 2.1e-2, median 1.1e-3, max 0.148
              # (the sampled PReLU slope model.6.weight -- one number with ~1e5-fold cancellation -- moved from 2.1e-2 to 0.118 when the 3x3
              #  kernel's summation order changed in round 5, every tensor-valued statistic stayed put; round 5 gave it a limit of its own, 0.3.
              #  Round 6: the three slopes and the maximum over the gradient norms are asserted over six weight seeds instead, see check())
              torch.float16: dict(out_atol=2e-2, grad_in_l2=0.16, grad_l2=0.22, grad_scalar='seeds', norm_median=1.2e-3, norm_max=None, buf_rtol=1e-3)},
              # f16 measured: out 9.4e-3, grad_in 8.0e-2 (0.92x stock f16), grads 0.109, median 5.8e-4, max 0.109
    'patchgan': {torch.float32: dict(out_atol=1.5e-5, grad_in_l2=5e-6, grad_l2=5e-6, norm_median=1e-7, norm_max=1e-6, buf_rtol=1e-6),
                 # f32 measured: out 5.8e-6, grad_in 1.8e-6, grads 2.3e-6, median 4.4e-8, max 3.5e-7, buffers 9.4e-8
                 torch.bfloat16: dict(out_atol=2.6e-2, grad_in_l2=0.17, grad_l2=0.23, norm_median=4e-3, norm_max=3.5e-2, buf_rtol=2e-4),
                 # bf16 measured: out 1.27e-2, grad_in 8.1e-2, grads 0.112 (the first conv's bias), median 1.8e-3, max 1.7e-2, buffers 9.6e-5
                 torch.float16: dict(out_atol=4e-3, grad_in_l2=6e-2, grad_l2=6.5e-2, norm_median=7e-4, norm_max=1.3e-2, buf_rtol=3e-5)},
                 # f16 measured: out 1.8e-3, grad_in 2.8e-2 (0.96 - 1.20x stock f16), grads 3.2e-2, median 3.5e-4, max 6.5e-3, buffers 1.1e-5
    # the two "next" networks: f32 / bf16 only (their f16 runs are not part of any BASELINE config)
    'label_net': {torch.float32: dict(out_atol=4e-5, grad_in_l2=4e-3, grad_l2=5e-3, norm_median=6e-5, norm_max=1.1e-3, buf_rtol=1e-5),
                  # f32 measured: out 1.7e-5, grad_in 1.9e-3, grads 2.4e-3, median 2.9e-5, max 5.3e-4, buffers 1.4e-7
                  torch.bfloat16: dict(out_atol=0.33, grad_in_l2=0.75, grad_l2=0.85, norm_median=3e-2, norm_max=0.17, buf_rtol=6e-4)},
                  # bf16 measured: out 0.164 (logits, |max| 2.8), grad_in 0.369, grads 0.422, median 1.45e-2, max 8.1e-2, buffers 3.0e-4
    'normal_net': {torch.float32: dict(out_atol=5e-5, grad_in_l2=2e-3, grad_l2=2e-3, norm_median=1e-5, norm_max=1e-4, buf_rtol=1e-5),
                   # f32 measured: out 2.1e-5, grad_in 9.0e-4, grads 7.9e-4, median 4.7e-6, max 4.1e-5, buffers 1.3e-7
                   torch.bfloat16: dict(out_atol=0.1, grad_in_l2=0.4, grad_l2=0.4, norm_median=5e-3, norm_max=4e-2, buf_rtol=4e-4)},
                   # bf16 measured: out 5.0e-2, grad_in 0.193, grads 0.204, median 2.4e-3, max 1.9e-2, buffers 1.8e-4
}
_STOCK = {}
_SEEDS = []


def _stock_seeds():
    if not _SEEDS:
        _SEEDS.append(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stock_lowp_seeds.npz')))
    return _SEEDS[0]


def _stock_lowp(kind, dtype):
    """input-gradient error (relative L2 against its own fp32 run) of a stock-PyTorch CPU evaluation of the ORACLE net with
    weights and activations in `dtype`, on the fixture inputs: what the storage precision alone costs"""
    key = (kind, dtype)
    if key in _STOCK:
        return _STOCK[key]
    if kind in ('g_a2b', 'g_b2a'):
        # the two generators: computed once on the host by tests/golden/make_stock_lowp.py (the same evaluation as below, weight seed 0) and
        # committed -- ~40 s of bf16 / f16 host convolutions per GPU test run until round 5
        gin = _stock_seeds()[f"gin.{'bf16' if dtype == torch.bfloat16 else 'f16'}.{kind}.0"]
        idx = (1, 2) if kind == 'g_a2b' else (0,)            # (RDFGenerator never reads its rgb argument: inputs 1 and 2 carry gradients)
        _STOCK[key] = {f'grad_in.{i}': float(v) for i, v in zip(idx, gin)}
        return _STOCK[key]
    from oracle import nets as onets
    mk, inputs, tag = {
        'g_a2b': (lambda: onets.RDFGenerator(), lambda: gen_inputs('g_a2b', 2, 60, 76), 'g_a2b_small'),
        'g_b2a': (lambda: onets.ResnetGenerator(1, 3, 64, 'IN2d'), lambda: (gen_inputs('g_b2a', 2, 60, 76)[1],), 'g_b2a_small'),
        'patchgan1': (lambda: onets.PatchGANDiscriminator(1), lambda: (det_tensor(3, 'disc1', (2, 1, 60, 76)),), 'd_a_small'),
        'patchgan3': (lambda: onets.PatchGANDiscriminator(3), lambda: (det_tensor(3, 'disc3', (2, 3, 60, 76)),), 'd_b_small'),
    }[kind]
    grads = {}
    for dt in (torch.float32, dtype):
        net = mk()
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
        net.train()
        net = net.to(dt)
        ins = [t.clone().to(dt).requires_grad_(True) for t in inputs()]
        o = net(*ins)
        o = o if isinstance(o, dict) else {'out': o}
        loss = 0
        for k, v in o.items():
            loss = loss + (v.float() * out_weights(tag, k, v.shape)).sum()
        loss.backward()
        grads[dt] = {i: t.grad.float() for i, t in enumerate(ins) if t.grad is not None}
    _STOCK[key] = {f'grad_in.{i}': float((grads[dtype][i] - grads[torch.float32][i]).norm() / grads[torch.float32][i].norm())
                   for i in grads[dtype]}
    return _STOCK[key]


DT = [torch.float32, torch.bfloat16, torch.float16]
IDS = ['f32', 'bf16', 'f16']


@pytest.mark.parametrize('dtype', DT, ids=IDS)
@pytest.mark.parametrize('nlspn,name', [(False, 'g_a2b_small'), (True, 'g_a2b_nlspn_small')])
def test_g_a2b(golden, nlspn, name, dtype):
    from rdfc_gan_amd.generator import RDFGenerator
    setup_dtype(dtype)
    net = RDFGenerator(pretrained_on_imagenet=False, use_nlspn_refine=nlspn, nlspn_configs=NLSPN_CFG)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to(DEV)
    rgb, depth, normal = (t.to(DEV) for t in gen_inputs('g_a2b', 2, 60, 76))
    res = run(net, (rgb, depth, normal), name)
    res.pop('grad_in.0', None)
    stock = _stock_lowp('g_a2b', dtype) if dtype != torch.float32 and not nlspn else None
    print(check(net, res, golden(name), stock=stock, **TOLS['g_a2b_nlspn' if nlspn else 'g_a2b'][dtype]))


@pytest.mark.parametrize('dtype', DT, ids=IDS)
def test_g_b2a(golden, dtype):
    from rdfc_gan_amd.generator import ResnetGenerator
    setup_dtype(dtype)
    net = ResnetGenerator(1, 3, 64, 'IN2d')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to(DEV)
    x = gen_inputs('g_b2a', 2, 60, 76)[1].to(DEV)
    stock = _stock_lowp('g_b2a', dtype) if dtype != torch.float32 else None
    print(check(net, run(net, (x,), 'g_b2a_small'), golden('g_b2a_small'), stock=stock, **TOLS['g_b2a'][dtype]))


@pytest.mark.parametrize('dtype', DT[1:], ids=IDS[1:])
@pytest.mark.parametrize('kind', ['g_a2b', 'g_b2a'])
def test_noisy_lowp_statistics_over_weight_seeds(kind, dtype):
    """round-5 verdict, weak item 2: the two statistics whose single-fixture limits were widened in round 5 -- the MAXIMUM relative error over
    the ~340 parameter-gradient norms, and the gradients of the PReLU slopes (one number each, heavy cancellation) -- as a statistic over SIX
    weight seeds with a limit tied to what 16-bit storage alone costs: a stock PyTorch evaluation of the oracle net with weights and
    activations in the same type (tests/golden/stock_lowp_seeds.npz, made on the host by tests/golden/make_stock_lowp.py; the reference of
    both sides is the oracle in fp32).  Over the seeds the stock evaluation's own maximum wanders 0.13 ... 0.32 (g_a2b, bf16) and
    0.07 ... 0.83 (g_b2a, bf16): no single draw bounds anything, the MEDIAN over the seeds does.
    Asserted: median over seeds of {max, median over the norms, worst PReLU slope} <= 1.5 x the stock evaluation's + a small floor."""
    from rdfc_gan_amd.generator import RDFGenerator, ResnetGenerator
    S = _stock_seeds()
    dn = 'bf16' if dtype == torch.bfloat16 else 'f16'
    names = [str(n) for n in S[f'names.{kind}']]
    nseed = len([k for k in S.files if k.startswith(f'ref.{kind}.')])
    assert nseed >= 5
    setup_dtype(dtype)
    mk = (lambda: RDFGenerator(pretrained_on_imagenet=False)) if kind == 'g_a2b' else (lambda: ResnetGenerator(1, 3, 64, 'IN2d'))
    tag = f'{kind}_small'
    ins = gen_inputs(kind, 2, 60, 76)
    ins = tuple(t.to(DEV) for t in (ins if kind == 'g_a2b' else (ins[1],)))
    rows = {'hip': [], 'stock': []}
    for seed in range(nseed):
        net = mk()
        net.load_state_dict(det_state_dict(net.state_dict(), seed))
        net.to(DEV).train()
        xs = [t.clone().requires_grad_(True) for t in ins]
        o = net(*xs)
        o = o if isinstance(o, dict) else {'out': o}
        loss = 0
        for k, v in o.items():
            loss = loss + (v * out_weights(tag, k, v.shape).to(DEV)).sum()
        loss.backward()
        got = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
        assert sorted(got) == names
        ref = S[f'ref.{kind}.{seed}']
        live = ref >= 1e-6 * ref.max()                     # (analytically-zero gradients: conv biases in front of an InstanceNorm)
        slopes = np.array([n.endswith('.weight') and dict(net.named_parameters())[n].numel() == 1 for n in names])
        for who, vec in (('hip', np.array([got[n] for n in names])), ('stock', S[f'{dn}.{kind}.{seed}'])):
            rel = np.abs(vec - ref) / np.maximum(ref, 1e-30)
            rows[who].append((float(rel[live].max()), float(np.median(rel[live])), float(rel[slopes & live].max()) if (slopes & live).any() else 0.0))
        del net
    hip_, stock = np.array(rows['hip']), np.array(rows['stock'])
    print(f'\n[seeds] {kind} {dn}: per seed (max | median | worst PReLU slope)\n  hip   ' + ' '.join(f'{a:.3f}|{b:.1e}|{c:.3f}' for a, b, c in hip_)
          + '\n  stock ' + ' '.join(f'{a:.3f}|{b:.1e}|{c:.3f}' for a, b, c in stock))
    mh, ms = np.median(hip_, axis=0), np.median(stock, axis=0)
    print(f'  medians over {nseed} seeds: hip {mh}, stock {ms}, ratio {mh / np.maximum(ms, 1e-12)}')
    for j, what in enumerate(('max over the gradient norms', 'median over the gradient norms', 'worst PReLU-slope gradient')):
        assert mh[j] <= 1.5 * ms[j] + (2e-2, 1e-3, 2e-2)[j], f'{kind} {dn}: {what}: median over seeds {mh[j]:.3e} against stock {ms[j]:.3e}'


@pytest.mark.parametrize('dtype', DT, ids=IDS)
@pytest.mark.parametrize('cin,name', [(1, 'd_a_small'), (3, 'd_b_small')])
def test_patchgan(golden, cin, name, dtype):
    from rdfc_gan_amd.discriminator import PatchGANDiscriminator
    setup_dtype(dtype)
    net = PatchGANDiscriminator(cin, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to(DEV)
    x = det_tensor(3, f'disc{cin}', (2, cin, 60, 76)).to(DEV)
    stock = _stock_lowp(f'patchgan{cin}', dtype) if dtype != torch.float32 else None
    print(check(net, run(net, (x,), name), golden(name), stock=stock, **TOLS['patchgan'][dtype]))


@pytest.mark.parametrize('dtype', DT, ids=IDS)
def test_g_a2b_full_resolution_odd_sizes(golden, dtype):
    """B=1, 228x304: 228->114->57->29->15 and the trailing-row crops of `_concat`."""
    from rdfc_gan_amd.generator import RDFGenerator
    setup_dtype(dtype)
    g = golden('g_a2b_full_b1')
    net = RDFGenerator(pretrained_on_imagenet=False)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to(DEV)
    rgb, depth, normal = (t.to(DEV) for t in gen_inputs('g_a2b', 1, 228, 304))
    res = run(net, (rgb, depth, normal), 'g_a2b_full_b1')
    res.pop('grad_in.0', None)
    print(check(net, res, g, **TOLS['g_a2b'][dtype]))
    # north-star acceptance metric: dense-depth RMSE (metres, x5+5 de-normalised) within 1e-3 of the reference's
    gt = (torch.from_numpy(g['eval.pred_depth']) * 0.9 + 0.02) * 5 + 5      # any fixed "ground truth"
    ours = res['eval.pred_depth'].cpu() * 5 + 5
    ref = torch.from_numpy(g['eval.pred_depth']) * 5 + 5
    rmse = lambda p: float(((p - gt) ** 2).mean().sqrt())
    assert abs(rmse(ours) - rmse(ref)) < 1e-3, (rmse(ours), rmse(ref))


@pytest.mark.parametrize('exact', ['exact', 'elided', 'batched'], ids=['exact_schedule', 'dead_forwards_elided', 'batched_calls'])
def test_train_step_against_upstream_trace(golden, exact):
    """Two optimisation steps of the HIP RDFCGAN (f32 mode) against the trace recorded from upstream's own
    RDFCGAN.optimize_parameters (tests/golden/step_trace.npz): upstream's literal schedule, the schedule without the two
    dead forwards, and the batched schedule (each network's two passes as one pass over the stacked inputs, BatchNorm
    statistics per half) -- the default of the benchmark."""
    batched, exact = exact == 'batched', exact == 'exact'
    import random
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(torch.float32)
    g = golden('step_trace')
    hp = STEP_HP
    cfg = dict(local_rank=0, lr=hp['lr'], beta1=hp['beta1'], beta2=hp['beta2'], gan_loss_type='lsgan', scheduler='linear',
               epoch=0, decay_epoch=100, n_epochs=150, lambda_A=hp['lambda_A'], lambda_B=hp['lambda_B'], lambda_L1=hp['lambda_L1'],
               model=dict(G_A2B=dict(type='RDFGenerator', pretrained_on_imagenet=False),
                          G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                          D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                          D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')))
    normals = {}
    m = RDFCGAN(torch.device(DEV), False, cfg, is_train=True, exact_schedule=exact,
                normal_provider=lambda tag, img: normals[tag], batch_calls=batched)
    assert m._batched() == batched
    for key, net in m.models.items():
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
    from rdfc_gan_amd import functional as Fh
    Fh.invalidate_packed_weights()
    m.train()
    for it in range(2):
        data, nA, nfA = step_batch(it, hp['B'], hp['H'], hp['W'])
        normals.update(real_A=nA.to(DEV), fake_A=nfA.to(DEV), fake_A_again=nfA.to(DEV))
        random.seed(100 + it)
        np.random.seed(100 + it)
        m.set_input(data)
        ls = m.optimize_parameters()
        extra = float(g[f'loss_extra.{it}'])
        for k, v in zip(g[f'loss_names.{it}'], g[f'loss_values.{it}']):
            ref = v - extra if k == 'loss_G' else v
            # step 0 is a pure function of the inputs; afterwards Adam (update ~ lr * g/|g| in its first steps)
            # turns rounding-level gradient differences into lr-sized weight differences
            lim = 2e-3 * abs(ref) + 1e-4 if it == 0 else 3e-2 * abs(ref) + 2e-3
            assert abs(ls[k] - ref) <= lim, f'step {it} {k}: {ls[k]} vs {ref}'
        atol = 5e-3 if it == 0 else 3e-2
        for got, ref in ((m.fake_B, g[f'fake_B.{it}']), (m.fake_A, g[f'fake_A.{it}'])):
            d = np.abs(got.detach().cpu().numpy() - ref)
            assert np.quantile(d, 0.999) < atol and d.max() < 10 * atol, (it, float(np.quantile(d, 0.999)), float(d.max()))
        for key, net in m.models.items():
            sd = net.state_dict()
            keep = [i for i, k in enumerate(g[f'sd_names.{key}']) if not noise_driven(key, str(k))]
            got = np.array([float(sd[k].double().norm()) for k in g[f'sd_names.{key}']])
            np.testing.assert_allclose(got[keep], g[f'sd_summary.{key}.{it}'][keep, 1], rtol=5e-3, atol=2e-4,
                                       err_msg=f'step {it} {key}')


def test_train_step_bf16_runs_and_tracks_f32():
    """bf16 (benchmark) configuration: finite losses close to the f32 parity mode on the same batch."""
    import random
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    from rdfc_gan_amd import functional as Fh
    hp = STEP_HP
    cfg = dict(local_rank=0, lr=hp['lr'], beta1=0.5, beta2=0.999, gan_loss_type='lsgan',
               model=dict(G_A2B=dict(type='RDFGenerator', pretrained_on_imagenet=False),
                          G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                          D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                          D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')))
    out = {}
    for dtype in (torch.float32, torch.bfloat16):
        setup_dtype(dtype)
        m = RDFCGAN(torch.device(DEV), False, cfg, is_train=True)
        for key, net in m.models.items():
            net.load_state_dict(det_state_dict(net.state_dict(), 0))
        Fh.invalidate_packed_weights()
        m.train()
        data, nA, _ = step_batch(0, hp['B'], hp['H'], hp['W'])
        data['gt_normal'] = nA
        random.seed(5)
        np.random.seed(5)
        m.set_input(data)
        out[dtype] = m.optimize_parameters()
    for k, v in out[torch.bfloat16].items():
        assert np.isfinite(v), k
        ref = out[torch.float32][k]
        assert abs(v - ref) <= 0.08 * abs(ref) + 0.02, f'{k}: bf16 {v} vs f32 {ref}'


def _step_cfg(hp):
    return dict(local_rank=0, lr=hp['lr'], beta1=0.5, beta2=0.999, gan_loss_type='lsgan',
                model=dict(G_A2B=dict(type='RDFGenerator', pretrained_on_imagenet=False),
                           G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                           D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                           D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['f32', 'bf16', 'f16'])
def test_wadain_against_golden(golden, dtype):
    """W-AdaIN (model_utils.py:72-90) alone against the fixture generated from the reference class: output,
    both input gradients, the style weight's and bias' gradients, at the four channel counts of the fusion layers."""
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.generator import AdaptiveInstanceNorm
    setup_dtype(dtype)
    g = golden('wadain')
    lim = {torch.float32: 2e-6, torch.bfloat16: 7e-3, torch.float16: 9e-4}[dtype]      # relative L2; measured 7.4e-7 / 3.6e-3 / 4.4e-4

    def rel(a, key):
        ref = torch.from_numpy(np.asarray(g[key])).double()
        return float((a.detach().double().cpu() - ref).norm() / (ref.norm() + 1e-30))

    for i, (c, h, w) in enumerate([(512, 5, 7), (768, 6, 9), (384, 9, 8), (192, 12, 10)], 1):
        m = AdaptiveInstanceNorm(c, c).to(DEV)
        m.load_state_dict(det_state_dict(m.state_dict(), i))
        Fh.invalidate_packed_weights()
        x = det_tensor(4, f'wadain_x{i}', (2, c, h, w)).to(DEV).requires_grad_(True)
        s_ = det_tensor(4, f'wadain_s{i}', (2, c, h, w)).to(DEV).requires_grad_(True)
        y = Fh.to_nchw(m(Fh.to_pm(x), Fh.to_pm(s_)))
        (y.float() * out_weights('wadain', str(i), y.shape).to(DEV)).sum().backward()
        gw = m.style.linear.weight_orig.grad
        errs = dict(y=rel(y, f'y{i}'), gx=rel(x.grad, f'gx{i}'), gs=rel(s_.grad, f'gs{i}'), gw=rel(gw[:32, :32], f'gw_corner{i}'),
                    gb=rel(m.style.linear.bias.grad, f'gb{i}'))
        assert all(v < lim for v in errs.values()), (i, c, errs)
        assert abs(float(gw.double().norm()) / g[f'gw_summary{i}'][1] - 1) < lim


def test_step_schedules_walk_the_same_trajectory(monkeypatch):
    """Where the step's independent pieces are ENQUEUED must not change what they compute: the discriminators' step behind G's
    backward (upstream's order), forked before it, or enqueued before it on the depth-branch stream (default); one stream or
    four.  Three f32 steps each; whatever differs comes from the weight gradients' float atomics, which Adam amplifies (the
    SAME schedule run twice drifts apart by the printed amount) -- a stream that read a buffer too early would be off by orders
    of magnitude more, or not finite."""
    import random
    from rdfc_gan_amd import rdfc_gan as rg
    from rdfc_gan_amd import functional as Fh
    hp = STEP_HP
    setup_dtype(torch.float32)

    def run(d_under_g, multi):
        monkeypatch.setattr(rg, '_D_UNDER_G', d_under_g)
        Fh.set_multi_stream(multi)
        m = rg.RDFCGAN(torch.device(DEV), False, _step_cfg(hp), is_train=True)
        for key, net in m.models.items():
            net.load_state_dict(det_state_dict(net.state_dict(), 0))
        Fh.invalidate_packed_weights()
        m.train()
        random.seed(5)
        np.random.seed(5)
        trace = []
        for it in range(3):
            data, nA, _ = step_batch(it % 2, hp['B'], hp['H'], hp['W'])
            data['gt_normal'] = nA
            m.set_input(data)
            trace.append(m.optimize_parameters())
        torch.cuda.synchronize()
        return trace

    try:
        ref = run(False, False)                              # upstream's order on one stream
        worst = {}
        for mode, multi in ((False, False), (False, True), (True, True), ('first', True), ('first', False)):
            got = run(mode, multi)
            dev = 0.0
            for it, (a, b) in enumerate(zip(ref, got)):
                assert list(a) == list(b)
                for k in a:
                    assert np.isfinite(b[k]), (mode, multi, it, k)
                    dev = max(dev, abs(a[k] - b[k]) / (abs(a[k]) + 0.05))
            worst[(mode, multi)] = dev
        print('max relative loss deviation from the reference schedule over 3 steps:', worst)
        noise = worst[(False, False)]                        # the same schedule twice: what the float atomics alone do
        for key, dev in worst.items():
            assert dev <= max(5 * noise, 5e-3), f'schedule {key}: {dev:.3e} against run-to-run noise {noise:.3e}'
    finally:
        Fh.set_multi_stream(True)


@pytest.mark.parametrize('graph_streams', ['one_stream'])     # (a fork / join capture of the four-stream step segfaulted in hipStreamEndCapture on ROCm 7.2; the option was removed in round 5)
def test_graph_mode_matches_eager(graph_streams):
    """RDFCGAN.enable_graph(): three captured HIP graphs (G step | D step | Adam) replayed per step must walk the
    same trajectory as the eager step (same host RNG draws, device-side Adam step count, BatchNorm batch counters); one-stream capture."""
    import random
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    hp = STEP_HP
    cfg = _step_cfg(hp)
    setup_dtype(torch.float32)
    runs = {}
    for mode in ('eager', 'graph'):
        m = RDFCGAN(torch.device(DEV), False, cfg, is_train=True)
        for key, net in m.models.items():
            net.load_state_dict(det_state_dict(net.state_dict(), 0))
        Fh.invalidate_packed_weights()
        m.train()
        if mode == 'graph':
            m.enable_graph(True)
        losses = []
        for it in range(3):
            data, nA, _ = step_batch(it, hp['B'], hp['H'], hp['W'])
            data['gt_normal'] = nA
            random.seed(40 + it)
            np.random.seed(40 + it)
            m.set_input(data)
            losses.append(m.optimize_parameters())
        sd = {k: {n: v.detach().float().cpu().clone() for n, v in net.state_dict().items()} for k, net in m.models.items()}
        runs[mode] = (losses, sd, [o.steps for o in m.optimizers.values()])
        if mode == 'graph':
            assert m._graph_mode and m._graphs is not None, 'capture fell back to eager launches'
            m.enable_graph(False)
    assert runs['eager'][2] == runs['graph'][2] == [3, 3]
    for it in range(3):
        for k, ref in runs['eager'][0][it].items():
            v = runs['graph'][0][it][k]
            # (both runs carry split-K atomic-order noise, which Adam's first lr-sized updates amplify step by step: 2.3e-3 seen at step 2)
            assert abs(v - ref) <= (2e-3 if it == 0 else 8e-3) * abs(ref) + 2e-4, f'step {it} {k}: graph {v} vs eager {ref}'
    worst = (0.0, '', '')
    for key in runs['eager'][1]:
        for n, ref in runs['eager'][1][key].items():
            got = runs['graph'][1][key][n]
            if n.endswith('num_batches_tracked'):
                assert int(got) == int(ref), (key, n, int(got), int(ref))
            else:
                # split-K atomics make the gradient sums order-dependent at rounding level, and Adam's first steps move
                # a weight by ~lr whatever the gradient's size: two eager runs differ by up to lr per step as well
                d = (got - ref).abs().max().item()
                # running statistics of the deepest layers are moments over a handful of values (1 x 1 ... 2 x 2 maps at this size): one
                # weight whose first Adam steps went the other way (a gradient at rounding level: +-lr either way) moves them by a few
                # per cent -- seen in 1 of 8 runs (round-4 tree: 4 of 8, other rounding) at 2.3 % of the largest entry
                rel = 5e-2 if n.endswith(('running_mean', 'running_var')) else 2e-2
                lim = 3.5 * hp['lr'] + rel * ref.abs().max().item() + 1e-4
                worst = max(worst, (d / lim, key, n))
                assert d <= lim or os.environ.get('RDFC_TEST_REPORT_ONLY'), (key, n, d)
    print('worst deviation / limit:', worst)


def test_graph_capture_after_eager_steps():
    """bench.py --graph captures AFTER an eager warm-up step: nothing of that step's autograd graph (its nodes are bound to the
    helper streams) may still be alive when the capture stream builds its own -- a leftover reference used to end in a
    segmentation fault inside hipStreamEndCapture."""
    import random
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    hp = STEP_HP
    setup_dtype(torch.bfloat16)
    m = RDFCGAN(torch.device(DEV), False, _step_cfg(hp), is_train=True)
    m.train()
    random.seed(3)
    np.random.seed(3)
    try:
        for it in range(4):
            if it == 2:
                m.enable_graph(True)
            data, nA, _ = step_batch(it % 2, hp['B'], hp['H'], hp['W'])
            data['gt_normal'] = nA
            m.set_input(data)
            losses = m.optimize_parameters()
            assert all(np.isfinite(v) for v in losses.values()), (it, losses)
        assert m._graph_mode and m._graphs is not None, 'capture fell back to eager launches'
    finally:
        m.enable_graph(False)


_EVAL_REF = {}


@pytest.mark.parametrize('dtype', DT, ids=IDS)
def test_eval_path_end_to_end_metrics_within_north_star_tolerance(dtype):
    """forward_test -> RDFGANMetric.evaluate_normalised (228x304) against the CPU oracle's generator + oracle.metrics on the same
    weights and batch, with a ground truth UNRELATED to the prediction (RMSE ~2.5 m: a bias of the prediction moves it one for
    one), three input seeds.  BASELINE.json's acceptance figure is "dense-depth RMSE within 1e-3 of reference":
      f32 (parity mode)           1e-4 m;
      f16 (the benchmarked mode)  1e-3 m -- measured <= 3.3e-4 over five weight seeds (tools/lowp_sim.py);
      bf16                        3e-3 m -- measured 1.2-1.5e-3.  Rounding the conv WEIGHTS to 8 significant bits is a fixed
        perturbation of the function: it shifts the mean prediction by ~3e-4 (normalised) = 1.5e-3 m whatever the activations
        are stored in (weights alone: 1.47e-3; activations alone: 9e-5), which is why the benchmark's default is f16."""
    from oracle import metrics as om
    from oracle import nets as on
    from rdfc_gan_amd.metrics import RDFGANMetric
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(dtype)
    m = RDFCGAN(torch.device(DEV), False, _step_cfg(STEP_HP), is_train=False)
    m.G_A2B.load_state_dict(det_state_dict(m.G_A2B.state_dict(), 0))
    from rdfc_gan_amd import functional as Fh
    Fh.invalidate_packed_weights()
    m.eval()
    lim = {torch.float32: 1e-4, torch.float16: 1e-3, torch.bfloat16: 3e-3}[dtype]
    worst = 0.0
    for tag in ('eval_e2e', 'eval_e2e_b', 'eval_e2e_c'):
        rgb, raw, normal = gen_inputs(tag, 2, 228, 304)
        gt = (raw * 0.9 + 0.05).clamp(-1, 1)                       # any fixed "ground truth" with holes where raw has them
        gt[raw <= -1.0] = -1.0
        out = m.forward_test(rgb=rgb, raw_depth=raw, gt_normal=normal)
        got = RDFGANMetric().evaluate_normalised(gt.to(DEV), out['pred_depth'], None).double().cpu().numpy()
        if tag not in _EVAL_REF:              # the oracle's side (one fp32 host forward per seed) is the same for the three dtype cases: once per session
            if 'net' not in _EVAL_REF:
                _EVAL_REF['net'] = on.RDFGenerator()
                _EVAL_REF['net'].load_state_dict(det_state_dict(_EVAL_REF['net'].state_dict(), 0))
                _EVAL_REF['net'].eval()
            with torch.no_grad():
                pred_ref = _EVAL_REF['net'](rgb, raw, normal)['pred_depth']
            _EVAL_REF[tag] = np.stack([om.image_metrics(om.denormalise(gt[k].numpy()), om.denormalise(pred_ref[k].numpy()), acc=np.float64) for k in range(2)])
        ref = _EVAL_REF[tag]
        worst = max(worst, float(np.abs(got[:, 0] - ref[:, 0]).max()))
        assert np.abs(got[:, 0] - ref[:, 0]).max() < lim, (tag, got[:, 0], ref[:, 0])        # RMSE, metres
        assert np.abs(got[:, 1] - ref[:, 1]).max() < lim, tag                                # MAE
        assert np.abs(got[:, 4:] - ref[:, 4:]).max() < (2e-4 if dtype == torch.float32 else 3e-3), tag   # REL, delta^1..3
    print(f'[eval path] {dtype}: worst |RMSE - reference RMSE| over 3 seeds x 2 images = {worst:.2e} m (limit {lim:.0e})')


# ---- full-size step: the HIP train step in every compute dtype against the oracle's step on the same inputs ------------------
_FULL = {}


def _oracle_full_step():
    """one oracle step (upstream's literal schedule, fp32, CPU) at B = 2, 228x304: read from tests/golden/oracle_steps.npz (made by
    tests/golden/make_oracle_steps.py on the host: the same call as below, ~1 min per GPU test run until round 5);
    RDFC_LIVE_ORACLE=1 recomputes it here"""
    if 'ref' in _FULL:
        return _FULL['ref']
    data, nA, nfA = step_batch(0, 2, 228, 304)
    normals = dict(real_A=nA, fake_A=nfA, fake_A_again=nfA)
    if os.environ.get('RDFC_LIVE_ORACLE') == '1':
        from make_oracle_steps import oracle_step
        losses, fB, fA, norms = oracle_step('full2')
    else:
        from make_oracle_steps import load
        losses, fB, fA, norms, sub, _sums = load('full2')
        assert sub == (1, 1)
    _FULL['ref'] = (data, normals, losses, fB, fA, norms)
    return _FULL['ref']


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['f32', 'bf16', 'f16'])
def test_full_size_step_against_oracle(dtype, monkeypatch):
    """B = 2 at the benchmark resolution (228x304: the odd-size crops, 57x76 / 59x78 G_B2A maps, 27x37 PatchGAN maps), one
    optimisation step, batched schedule: every loss entry, fake_B / fake_A and the norm of every updated tensor against the
    oracle's step (upstream's literal schedule on the CPU).  f32 pins the arithmetic; bf16 / f16 bound what storage
    precision costs at full size (measured values in the comments; limits <= 2x measured)."""
    from rdfc_gan_amd import rdfc_gan as rg
    data, normals, ref_losses, ref_fB, ref_fA, ref_norms = _oracle_full_step()
    setup_dtype(dtype)
    monkeypatch.setattr(rg, 'draw_pepper_mask', lambda shape, snr=0.98, p=0.9: None)        # no noise on either side
    cfg = _step_cfg(STEP_HP)
    dn = {k: v.to(DEV) for k, v in normals.items()}
    m = rg.RDFCGAN(torch.device(DEV), False, cfg, is_train=True, normal_provider=lambda tag, img: dn[tag], batch_calls=True)
    for key, net in m.models.items():
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
    from rdfc_gan_amd import functional as Fh
    Fh.invalidate_packed_weights()
    m.train()
    m.set_input(data)
    ls = m.optimize_parameters()
    #                    losses (rel)   fake maps (q99.9 abs, max abs)   weight norms (rel)
    # measured (f32 | bf16 | f16): losses 2.8e-7 | 3.0e-3 | 1.7e-4; fake maps q99.9 7.2e-6 | 9.0e-2 | 1.1e-2, max 1.3e-5 | 0.147 | 1.7e-2;
    # weight norms 5.4e-4 | 5.4e-3 | 3.2e-3 (one Adam step moves a weight by lr whatever its gradient's size: sign flips of near-zero gradients)
    lim = {torch.float32: (1e-5, 5e-5, 1e-4, 1.1e-3), torch.bfloat16: (6e-3, 0.18, 0.3, 1.1e-2), torch.float16: (4e-4, 2.3e-2, 3.5e-2, 6.5e-3)}[dtype]
    report, bad = [], []
    for k, v in ls.items():
        ref = ref_losses[k]
        report.append(f'{k}: {abs(v - ref) / (abs(ref) + 1e-6):.2e}')
        if not abs(v - ref) <= lim[0] * abs(ref) + 1e-4:
            bad.append(f'{k}: {v} vs {ref}')
    for name, got, ref in (('fake_B', m.fake_B, ref_fB), ('fake_A', m.fake_A, ref_fA)):
        d = (got.detach().float().cpu() - ref).abs()
        q = float(torch.quantile(d.flatten()[::7], 0.999))
        report.append(f'{name}: q99.9 {q:.2e} max {float(d.max()):.2e}')
        if not (q < lim[1] and float(d.max()) < lim[2]):
            bad.append(f'{name}: q99.9 {q} max {float(d.max())}')
    worst = 0.0
    for key, net in m.models.items():
        for n, v in net.state_dict().items():
            if v.dtype.is_floating_point and not noise_driven(key, n) and not n.endswith('num_batches_tracked'):
                r = ref_norms[key][n]
                worst = max(worst, abs(float(v.double().norm()) - r) / (abs(r) + 1e-4))
    report.append(f'weight norms: worst rel {worst:.2e}')
    print('[full-size step]', '; '.join(report))
    if not worst < lim[3]:
        bad.append(f'weight norms: worst rel {worst}')
    assert not bad, '; '.join(bad)
