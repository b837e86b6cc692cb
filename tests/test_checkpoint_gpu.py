"""GPU: checkpoints in upstream's format (lib/utils/checkpoint.py:21-58: '{name}_state_dict', '{name}_optimizer_state_dict' holding
torch.optim.Adam.state_dict(), '{name}_lr_scheduler_state_dict') -- written by this package and read back, and written the way
upstream writes them (plain torch modules + torch.optim.Adam) and resumed here."""
import itertools
import random

import numpy as np
import pytest
import torch

from cases import STEP_HP, step_batch
from detweights import det_state_dict
from test_modules_gpu import DEV, _step_cfg, setup_dtype

pytestmark = pytest.mark.gpu


def _model():
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    cfg = _step_cfg(STEP_HP)
    cfg.update(scheduler='linear', epoch=0, decay_epoch=100, n_epochs=150)
    return RDFCGAN(torch.device(DEV), False, cfg, is_train=True)


def _step(m, it):
    data, nA, _ = step_batch(it, STEP_HP['B'], STEP_HP['H'], STEP_HP['W'])
    data['gt_normal'] = nA
    random.seed(7 + it)
    np.random.seed(7 + it)
    m.set_input(data)
    return m.optimize_parameters()


def test_save_resume_round_trip_continues_identically(tmp_path):
    """one step, save, one more step; a fresh model resumed from the file takes that second step to the same weights"""
    from rdfc_gan_amd import functional as Fh
    setup_dtype(torch.float32)
    a = _model()
    for key, net in a.models.items():
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
    Fh.invalidate_packed_weights()
    a.train()
    _step(a, 0)
    path = str(tmp_path / 'ckpt.pth')
    a.save_ckpt(path, meta=dict(epoch=3))
    ck = torch.load(path, map_location='cpu')
    assert {'G_A2B_state_dict', 'G_B2A_state_dict', 'disc_A_state_dict', 'disc_B_state_dict', 'G_optimizer_state_dict',
            'D_optimizer_state_dict', 'G_lr_scheduler_state_dict', 'D_lr_scheduler_state_dict', 'meta'} <= set(ck)
    st = ck['G_optimizer_state_dict']['state'][0]
    assert set(st) == {'step', 'exp_avg', 'exp_avg_sq'} and float(st['step']) == 1.0          # torch.optim.Adam's layout
    la = _step(a, 1)
    b = _model()
    assert b.resume_from(path) == 3
    b.train()
    lb = _step(b, 1)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(la[k]) + 1e-6, k
    for key in a.models:
        sa, sb = a.models[key].state_dict(), b.models[key].state_dict()
        for k in sa:
            if sa[k].dtype.is_floating_point:
                assert float((sa[k] - sb[k]).abs().max()) <= 2e-5 * (float(sa[k].abs().max()) + 1e-3), (key, k)    # split-K atomics: order noise


def test_resume_from_a_checkpoint_written_upstreams_way(tmp_path):
    """state written from plain torch modules with torch.optim.Adam (what upstream's save_checkpoint stores) resumes here:
    same weights, same Adam moments, same step count"""
    from oracle import nets as on
    setup_dtype(torch.float32)
    m = _model()
    net = on.RDFGenerator()
    net.load_state_dict(det_state_dict(net.state_dict(), 1))
    assert list(net.state_dict()) == list(m.G_A2B.state_dict())
    opt = torch.optim.Adam(itertools.chain(net.parameters()), lr=2e-3, betas=(0.5, 0.999))
    for p in net.parameters():
        p.grad = torch.randn_like(p) * 0.01
    opt.step()
    # upstream's optimizer_G spans G_A2B and G_B2A: build the combined optimizer state the same way
    g_b2a = {k: v.detach().cpu().clone() for k, v in m.G_B2A.state_dict().items()}
    n_a2b = len(list(net.parameters()))
    state = {i: {k: v.clone() for k, v in opt.state_dict()['state'][i].items()} for i in range(n_a2b)}
    for j, p in enumerate(m.G_B2A.parameters()):
        state[n_a2b + j] = dict(step=torch.tensor(1.0), exp_avg=torch.zeros_like(p, device='cpu'), exp_avg_sq=torch.zeros_like(p, device='cpu'))
    groups = [dict(opt.state_dict()['param_groups'][0], params=list(range(n_a2b + len(list(m.G_B2A.parameters())))))]
    path = str(tmp_path / 'upstream.pth')
    torch.save(dict(meta=dict(epoch=5), G_A2B_state_dict={'module.' + k: v for k, v in net.state_dict().items()},   # DDP prefix
                    G_B2A_state_dict=g_b2a, G_optimizer_state_dict=dict(state=state, param_groups=groups)), path)
    assert m.resume_from(path) == 5
    for k, v in net.state_dict().items():
        assert torch.equal(m.G_A2B.state_dict()[k].cpu(), v), k
    got = m.optimizer_G.state_dict()['state']
    for i in range(n_a2b):
        assert torch.equal(got[i]['exp_avg'].cpu(), state[i]['exp_avg']) and torch.equal(got[i]['exp_avg_sq'].cpu(), state[i]['exp_avg_sq'])
    assert m.optimizer_G.steps == 1


def test_f16_overflow_skips_the_whole_step_and_resume_keeps_the_scaler_state(tmp_path):
    """Model-level overflow path (the kernels are covered in test_kernels_gpu.py): with an absurd loss scale the f16 backward overflows,
    BOTH optimizers must skip -- parameters, Adam moments and the step count that a checkpoint stores unchanged --, the scale halves, and a
    model resumed from a checkpoint continues with that scale and those skip counters (not with the initial 1024 / 0)."""
    from rdfc_gan_amd import functional as Fh
    setup_dtype(torch.float16)
    try:
        a = _model()
        for key, net in a.models.items():
            net.load_state_dict(det_state_dict(net.state_dict(), 0))
        Fh.invalidate_packed_weights()
        a.train()
        assert a.scaler is not None and a.scaler.dynamic
        _step(a, 0)                                                   # a clean step first: moments are non-zero afterwards
        assert a.scaler.skipped() == [0, 0]
        before = {n: (o.flat.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone()) for n, o in a.optimizers.items()}
        a.scaler.set_scale(2.0 ** 40)                                 # every gradient overflows half precision
        losses = _step(a, 1)
        assert all(np.isfinite(v) for v in losses.values())           # the forward (and the logged losses) are unaffected
        assert a.scaler.skipped() == [1, 1]
        assert a.scaler.scale() == 2.0 ** 39                          # backed off once
        for n, o in a.optimizers.items():
            for got, ref in zip((o.flat, o.exp_avg, o.exp_avg_sq), before[n]):
                assert torch.equal(got, ref), f'{n}: a skipped step changed optimizer state'
            assert o.steps_taken() == 1
            assert float(o.state_dict()['state'][0]['step']) == 1.0   # what a checkpoint stores: steps really taken
        a.scaler.set_scale(512.0)
        path = str(tmp_path / 'f16.pth')
        a.save_ckpt(path, meta=dict(epoch=1))
        assert torch.load(path, map_location='cpu')['loss_scaler_state_dict'] == dict(scale=512.0, clean_steps=0, skipped=[1, 1])
        la = _step(a, 2)
        b = _model()
        assert b.scaler.scale() == 1024.0
        b.resume_from(path)
        assert b.scaler.scale() == 512.0 and b.scaler.skipped() == [1, 1]
        assert all(o.steps_taken() == 1 for o in b.optimizers.values())
        b.train()
        lb = _step(b, 2)
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-3 * abs(la[k]) + 1e-4, (k, la[k], lb[k])
    finally:
        setup_dtype(torch.float32)
