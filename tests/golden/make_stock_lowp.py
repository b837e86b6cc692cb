"""Generates tests/golden/stock_lowp_seeds.npz: what 16-bit STORAGE alone does to the gradients of the generators, over several
weight seeds -- the yardstick for the two noisy bf16 / f16 statistics of tests/test_modules_gpu.py (round-5 verdict, weak item 2: a
single-fixture maximum that moves 5x with the summation order measures nothing).

For every net (g_a2b = RDFGenerator, g_b2a = ResnetGenerator) and weight seed 0 .. NSEED-1 (tests/golden/detweights.py), on the fixture
inputs of tests/golden/cases.py, train mode, the fixed random cotangents of the module tests:

    ref.<net>.<seed>        parameter-gradient norms of the ORACLE net (oracle/nets.py, pinned bit-equal to the reference's classes by
                            tests/test_oracle_golden.py) in fp32 on the host
    <dt>.<net>.<seed>       the same from a STOCK PyTorch evaluation with weights and activations stored in dt = bf16 | f16
    gin.<dt>.<net>.<seed>   relative L2 error of that evaluation's input gradients against the fp32 one (what _stock_lowp measured at run
                            time on every GPU test run until round 5: ~40 s of host convolutions in bf16)
    names.<net>             parameter names (row order of the vectors above)

Nothing of the reference is needed (the oracle is pinned separately); runs on the CPU in ~3 minutes:

    python tests/golden/make_stock_lowp.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from cases import gen_inputs, out_weights  # noqa: E402
from detweights import det_state_dict  # noqa: E402

NSEED = 6
NETS = {
    'g_a2b': (lambda on: on.RDFGenerator(), lambda: gen_inputs('g_a2b', 2, 60, 76), 'g_a2b_small'),
    'g_b2a': (lambda on: on.ResnetGenerator(1, 3, 64, 'IN2d'), lambda: (gen_inputs('g_b2a', 2, 60, 76)[1],), 'g_b2a_small'),
}
DTS = {'bf16': torch.bfloat16, 'f16': torch.float16}


def evaluate(mk, inputs, tag, seed, dt):
    """-> ({parameter: gradient norm}, {input index: gradient}) of one train-mode forward + backward with storage type dt"""
    from oracle import nets as onets
    net = mk(onets)
    net.load_state_dict(det_state_dict(net.state_dict(), seed))
    net.train()
    net = net.to(dt)
    ins = [t.clone().to(dt).requires_grad_(True) for t in inputs()]
    o = net(*ins)
    o = o if isinstance(o, dict) else {'out': o}
    loss = 0
    for k, v in o.items():
        loss = loss + (v.float() * out_weights(tag, k, v.shape)).sum()
    loss.backward()
    norms = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
    gin = {i: t.grad.float() for i, t in enumerate(ins) if t.grad is not None}
    return norms, gin


def main():
    out = {}
    for kind, (mk, inputs, tag) in NETS.items():
        names = None
        for seed in range(NSEED):
            ref, gref = evaluate(mk, inputs, tag, seed, torch.float32)
            if names is None:
                names = sorted(ref)
                out[f'names.{kind}'] = np.array(names)
            out[f'ref.{kind}.{seed}'] = np.array([ref[n] for n in names])
            for dn, dt in DTS.items():
                low, glow = evaluate(mk, inputs, tag, seed, dt)
                out[f'{dn}.{kind}.{seed}'] = np.array([low[n] for n in names])
                out[f'gin.{dn}.{kind}.{seed}'] = np.array([float((glow[i] - gref[i]).norm() / gref[i].norm()) for i in sorted(gref)])
                rel = np.abs(out[f'{dn}.{kind}.{seed}'] - out[f'ref.{kind}.{seed}']) / np.maximum(out[f'ref.{kind}.{seed}'], 1e-30)
                live = out[f'ref.{kind}.{seed}'] >= 1e-6 * out[f'ref.{kind}.{seed}'].max()
                print(f'{kind} seed {seed} {dn}: grad_in {out[f"gin.{dn}.{kind}.{seed}"]}, param-grad norms median {np.median(rel[live]):.2e} '
                      f'max {rel[live].max():.2e}', flush=True)
    np.savez_compressed(os.path.join(HERE, 'stock_lowp_seeds.npz'), **out)


if __name__ == '__main__':
    torch.manual_seed(0)
    main()
