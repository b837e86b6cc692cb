"""GPU: the BASELINE.json configurations at their OWN shapes (the module / step parity tests run at 60x76 and B = 2 at 228x304):
  * configs[4]: 480x640, batch 4, f16 activations with f32 loss accumulation -- one train step + the eval forward against the oracle;
  * configs[1]: batch 8 (16 stacked samples per pass) at 228x304 -- the batched schedule against the unbatched one, BatchNorm
    buffers included, and a bf16 / f16 step;
  * configs[2]: NLSPN with 18 propagation steps at 228x304 against oracle.nlspn;
  * the per-pixel loss kernel against the reference's own values (tests/golden/losses.npz);
  * build_model() with the dict train.py:87-103 assembles from upstream's YAML."""

import os

import numpy as np
import pytest
import torch

from cases import NLSPN_CFG, gen_inputs, step_batch
from detweights import det_state_dict, det_tensor

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def setup_dtype(dtype):
    from rdfc_gan_amd import functional as Fh
    Fh.set_compute_dtype(dtype)
    Fh.invalidate_packed_weights()


def _cfg(lr=2e-3):
    return dict(local_rank=0, lr=lr, beta1=0.5, beta2=0.999, gan_loss_type='lsgan',
                model=dict(G_A2B=dict(type='RDFGenerator', pretrained_on_imagenet=False),
                           G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                           D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                           D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')))


def _load_det(model):
    from rdfc_gan_amd import functional as Fh
    for name, net in model.models.items():
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
    Fh.invalidate_packed_weights()


def _one_step(model, B, H, W, seed=0):
    import random
    random.seed(seed)
    np.random.seed(seed)
    data, nA, _ = step_batch(0, B, H, W)
    data['gt_normal'] = nA
    model.set_input(data)
    return model.optimize_parameters()


def test_config5_480x640_batch4_f16_step_and_eval_forward():
    """SUN RGB-D shape of BASELINE configs[4] (both sides divisible by 16: no crop path; 30x40 ... 480x640 maps): one optimisation
    step in f16 with dynamic loss scaling -- finite losses, no skipped update, gradients exchanged at the initial scale -- and
    G_A2B's eval forward at this size against the CPU oracle on the same weights (B = 1)."""
    from oracle import nets as on
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(torch.float16)
    m = RDFCGAN(torch.device(DEV), False, _cfg(), is_train=True)
    _load_det(m)
    m.train()
    losses = _one_step(m, 4, 480, 640)
    assert len(losses) == 13 and all(np.isfinite(v) for v in losses.values()), losses
    assert m.scaler is not None and m.scaler.skipped() == [0, 0] and m.scaler.scale() == 1024.0
    for o in m.optimizers.values():
        assert bool(torch.isfinite(o.flat).all())
    # eval forward at 480x640 against the oracle (outputs in [-1, 1]): the f32 parity mode pins the kernels at this size, f16 is
    # the configuration's storage type (CPU simulation of f16 storage, tools/lowp_sim.py at this size: 5.7e-3 max, 7.4e-4 rms)
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.generator import RDFGenerator
    ref = on.RDFGenerator()
    ref.load_state_dict(det_state_dict(ref.state_dict(), 0))
    ref.eval()
    rgb, raw, normal = gen_inputs('cfg5', 1, 480, 640)
    with torch.no_grad():
        exp = ref(rgb, raw, normal)
    # ... first through the modules that have just trained (upstream's validation flow: load weights, eval(), forward)
    m.G_A2B.load_state_dict(det_state_dict(m.G_A2B.state_dict(), 0))
    Fh.invalidate_packed_weights()
    m.eval()
    got = m(rgb=rgb, raw_depth=raw, gt_normal=normal)
    e = (got['pred_depth'].float().cpu() - exp['pred_depth']).abs()
    print(f'[480x640 eval] f16, trained modules reloaded: pred_depth max {float(e.max()):.2e} rms {float(e.pow(2).mean().sqrt()):.2e}')
    assert float(e.max()) < 2e-2 and float(e.pow(2).mean().sqrt()) < 2e-3
    del m, got
    torch.cuda.empty_cache()
    for dtype, lim_max, lim_rms in ((torch.float32, 1e-4, 1e-5), (torch.float16, 2e-2, 2e-3)):
        setup_dtype(dtype)
        net = RDFGenerator(pretrained_on_imagenet=False)
        net.load_state_dict(det_state_dict(net.state_dict(), 0))
        net.to(DEV).eval()
        with torch.no_grad():
            got = net(rgb.to(DEV), raw.to(DEV), normal.to(DEV))
        for k in ('pred_depth', 'depth_map_1', 'depth_map_2', 'confidence_map_1', 'confidence_map_2'):
            e = (got[k].float().cpu() - exp[k]).abs()
            print(f'[480x640 eval] {dtype} {k}: max {float(e.max()):.2e} rms {float(e.pow(2).mean().sqrt()):.2e}')
            assert float(e.max()) < lim_max and float(e.pow(2).mean().sqrt()) < lim_rms, (dtype, k, float(e.max()))


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['f16', 'bf16'])
def test_eval_forward_480x640_is_reproducible_run_to_run(dtype):
    """Regression (round 3): with the two branches of G_A2B on two streams and 480x640 maps (more workgroups than the chip holds,
    two kernels sharing every CU's LDS queue) the LDS-DMA kernels returned rare wrong half-tiles, different from run to run: a ring
    slot was refilled while a late fragment read of the previous step was still queued (common.h: wait_vm_and_reads).  The forward
    has no atomics: ten runs on the same weights and inputs must agree bit for bit."""
    from rdfc_gan_amd.generator import RDFGenerator
    setup_dtype(dtype)
    net = RDFGenerator(pretrained_on_imagenet=False)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to(DEV).eval()
    inputs = [t.to(DEV) for t in gen_inputs('cfg5', 1, 480, 640)]
    first = None
    for r in range(10):
        with torch.no_grad():
            out = {k: v.clone() for k, v in net(*inputs).items()}
        torch.cuda.synchronize()
        if first is None:
            first = out
            continue
        for k in out:
            assert torch.equal(out[k], first[k]), (r, k, float((out[k].float() - first[k].float()).abs().max()))


_F32_STEPS = {}


def _two_steps(dtype):
    """losses of two optimisation steps at the benchmark's launch sizes (batch 8 at 228x304, fixed weights and batches)"""
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(dtype)
    m = RDFCGAN(torch.device(DEV), False, _cfg(), is_train=True)
    _load_det(m)
    m.train()
    assert m._batched()
    out = [_one_step(m, 8, 228, 304, seed=it) for it in range(2)]
    for o in m.optimizers.values():
        assert bool(torch.isfinite(o.flat).all())
    if m.scaler is not None:
        assert m.scaler.skipped() == [0, 0]
    del m
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_config2_batch8_steps_follow_the_f32_mode(dtype):
    """the benchmark's launch sizes (16 stacked samples per pass at 228x304) in both 16-bit modes: two steps with finite losses
    and weights, no skipped update -- and every loss within the storage format's reach of the f32 parity mode's on the same
    weights and batches (step 1: the forward at this size; step 2: after one backward + Adam update at this size)."""
    if 'f32' not in _F32_STEPS:
        _F32_STEPS['f32'] = _two_steps(torch.float32)
    ref, got = _F32_STEPS['f32'], _two_steps(dtype)
    rtol = {torch.float16: (1e-3, 1.5e-2), torch.bfloat16: (1e-2, 4e-2)}[dtype]          # measured: 1.3e-4, 3.0e-3 | 2.2e-3, 1.0e-2
    for it in range(2):
        assert all(np.isfinite(v) for v in got[it].values()), (it, got[it])
        dev = {k: abs(got[it][k] - ref[it][k]) / max(abs(ref[it][k]), 1e-2) for k in ref[it]}
        worst = max(dev, key=dev.get)
        print(f'[batch 8, {dtype}] step {it + 1}: worst loss deviation from the f32 mode {dev[worst]:.2e} ({worst}: {got[it][worst]:.5f} vs {ref[it][worst]:.5f})')
        assert dev[worst] <= rtol[it], (it, worst, got[it][worst], ref[it][worst])


def test_config2_batch8_batched_schedule_equals_unbatched_f32():
    """At batch 8 the batched schedule (each network once over [real | fake] stacked, BatchNorm statistics per half) must leave
    behind what upstream's call-by-call schedule does: same losses, same BatchNorm running statistics and batch counts, same
    updated weights (f32 mode; the two differ in launch shapes and summation order only)."""
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(torch.float32)
    out = {}
    for batched in (True, False):
        m = RDFCGAN(torch.device(DEV), False, _cfg(), is_train=True, batch_calls=batched)
        _load_det(m)
        m.train()
        assert m._batched() == batched
        losses = _one_step(m, 8, 228, 304)
        bufs = {f'{n}.{k}': v.detach().double().cpu() for n, net in m.models.items() for k, v in net.named_buffers()}
        wts = {n: o.flat.detach().double().cpu() for n, o in m.optimizers.items()}
        out[batched] = (losses, bufs, wts)
        del m
        torch.cuda.empty_cache()
    (la, ba, wa), (lb, bb, wb) = out[True], out[False]
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-5 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    assert ba.keys() == bb.keys() and any(k.endswith('num_batches_tracked') for k in ba)
    for k in ba:
        if k.endswith('num_batches_tracked'):
            assert float(ba[k]) == float(bb[k]), k
        else:
            assert float((ba[k] - bb[k]).abs().max()) <= 1e-5 * float(bb[k].abs().max() + 1e-3), k
    for n in wa:         # one Adam step moves every weight by ~lr whatever its gradient: compare the update directions in bulk
        d = (wa[n] - wb[n]).abs()
        # (entries whose gradient is rounding noise -- conv biases in front of an InstanceNorm -- take +-lr steps of either sign)
        assert float(d.mean()) < 5e-5 and float((d > 1e-3).double().mean()) < 1e-2, (n, float(d.mean()), float((d > 1e-3).double().mean()))


def test_config3_batch8_step_against_oracle(monkeypatch):
    """BASELINE configs[2] at its own batch: ONE optimisation step of the four-network schedule with NLSPN on (18 propagation steps, TGASS,
    confidence propagation) at batch 8, 228x304, f32 parity mode, against oracle.step.HotPathStep running upstream's literal schedule on
    the host with the oracle's NLSPN (round-4 verdict: config #3 had only been compared at B = 1 / through the bench).  Every loss entry,
    fake_B / fake_A and the norms of the updated weights; limits are 2-3x what MI355X measured (in the comment at the assertion)."""
    from rdfc_gan_amd import rdfc_gan as rg
    from cases import noise_driven
    setup_dtype(torch.float32)
    B, H, W = 8, 228, 304
    data, nA, nfA = step_batch(0, B, H, W)
    normals = dict(real_A=nA, fake_A=nfA, fake_A_again=nfA)
    # the oracle's side of the comparison is 130 s of host time (128 threads) per run: read from tests/golden/oracle_steps.npz, which
    # tests/golden/make_oracle_steps.py computes with the very call below (RDFC_LIVE_ORACLE=1: recompute here).  The file keeps every 2nd /
    # 3rd pixel per axis of fake_B / fake_A and per-image sums + sums of squares of the full maps.
    if os.environ.get('RDFC_LIVE_ORACLE') == '1':
        from make_oracle_steps import oracle_step
        ref, ref_fB, ref_fA, ref_norms = oracle_step('c3b8')
        sB = sA = 1
        ref_sums = None
    else:
        from make_oracle_steps import load
        ref, ref_fB, ref_fA, ref_norms, (sB, sA), ref_sums = load('c3b8')
    monkeypatch.setattr(rg, 'draw_pepper_mask', lambda shape, snr=0.98, p=0.9: None)        # no noise on either side
    cfg = _cfg()
    cfg.update(lambda_A=100.0, lambda_B=100.0, lambda_L1=100.0)
    cfg['model']['G_A2B'].update(use_nlspn_refine=True, nlspn_configs=NLSPN_CFG)
    dn = {k: v.to(DEV) for k, v in normals.items()}
    m = rg.RDFCGAN(torch.device(DEV), False, cfg, is_train=True, normal_provider=lambda tag, img: dn[tag], batch_calls=True)
    _load_det(m)
    m.train()
    m.set_input(data)
    ls = m.optimize_parameters()
    worst_l = max(abs(ls[k] - ref[k]) / (abs(ref[k]) + 1e-6) for k in ls)
    got_B, got_A = m.fake_B.detach().float().cpu(), m.fake_A.detach().float().cpu()
    dB = (got_B[:, :, ::sB, ::sB] - ref_fB).abs()
    dA = (got_A[:, :, ::sA, ::sA] - ref_fA).abs()
    if ref_sums is not None:            # every pixel of the full maps, through per-image sums and sums of squares in f64
        for nm, t in (('fake_B', got_B), ('fake_A', got_A)):
            d = t.double().flatten(1)
            s12 = torch.stack([d.sum(1), (d * d).sum(1)], 1).numpy()
            npx = d.shape[1]
            assert np.abs(s12[:, 0] - ref_sums[nm][:, 0]).max() / npx < 1e-5 and np.abs(s12[:, 1] - ref_sums[nm][:, 1]).max() / npx < 1e-5, (nm, s12, ref_sums[nm])
    worst_n = 0.0
    for key, net in m.models.items():
        for n, v in net.state_dict().items():
            if v.dtype.is_floating_point and not noise_driven(key, n) and not n.endswith('num_batches_tracked'):
                r = ref_norms[key][n]
                worst_n = max(worst_n, abs(float(v.double().norm()) - r) / (abs(r) + 1e-4))
    print(f'[config 3, batch 8, f32] losses worst rel {worst_l:.2e}; fake_B max {float(dB.max()):.2e} fake_A max {float(dA.max()):.2e}; weight norms worst rel {worst_n:.2e}')
    # measured on MI355X (round 5): losses 9.4e-8, fake_B 1.05e-4, fake_A 1.5e-5, weight norms 6.3e-4 (Adam's first lr-sized step amplifies the
    # f32 summation-order differences of the gradients: the norm statistic is the loosest of the four)
    assert worst_l < 2e-6 and float(dB.max()) < 3e-4 and float(dA.max()) < 5e-5 and worst_n < 1.5e-3, (worst_l, float(dB.max()), float(dA.max()), worst_n)


def test_config3_nlspn_18_steps_228x304_against_oracle():
    """the shipped NLSPN configuration (prop_time 18, TGASS, confidence propagation) at the benchmark resolution, f32, against
    oracle.nlspn on the same guidance-conv weights: output, and the gradients of the initial map / confidence / guidance"""
    from oracle.nlspn import NLSPNProp
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.nlspn import NLPSN
    setup_dtype(torch.float32)
    B, H, W, T = 1, 228, 304, NLSPN_CFG['prop_time']
    # A well-posed instance: the initial map is a depth map (smooth, with rectangular holes), the guidance conv's weights are
    # small (offsets up to ~0.2 px, |affinity| sums up to ~1.2).  With weights of 0.02 and more or a white-noise map the 18-step
    # recursion amplifies rounding differences by orders of magnitude (the oracle evaluated in f32 and in f64 differs by 2e-3
    # .. 0.7 there); here f32 and f64 oracle agree to 7e-6 while the propagation still moves pixels by up to 0.6.
    _, raw, _ = gen_inputs('nlspn18', B, H, W)
    guide = det_tensor(61, 'guide', (B, 8, H, W))
    init = raw.clone()
    conf = torch.sigmoid(det_tensor(61, 'conf', (B, 1, H, W)))
    gout = det_tensor(61, 'gout', (B, 1, H, W))
    w, b = det_tensor(61, 'w', (24, 8, 3, 3)) * 0.005, det_tensor(61, 'b', (24,)) * 0.005
    ref = NLSPNProp(8, 3, T, 'TGASS', 0.5, conf_prop=True)
    m = NLPSN(8, 1, 3, 3, T, 'TGASS', 0.5, True, False).to(DEV)
    with torch.no_grad():
        for mod in (ref, m):
            mod.conv_offset_aff.weight.copy_(w)
            mod.conv_offset_aff.bias.copy_(b)
    Fh.invalidate_packed_weights()
    g_r, i_r, c_r = (t.clone().requires_grad_(True) for t in (guide, init, conf))
    y_r, _, _ = ref(i_r, g_r, c_r)
    y_r.backward(gout)
    g_d = guide.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    i_d, c_d = init.to(DEV).requires_grad_(True), conf.to(DEV).requires_grad_(True)
    assert m._fused()
    y_d, feats, _, _, _ = m(i_d, g_d, c_d)
    assert len(feats) == T
    y_d.backward(gout.to(DEV))

    def rel(a, b_):
        a, b_ = a.double().cpu(), b_.double()
        return float((a - b_).norm() / (b_.norm() + 1e-30))
    e_out = float((y_d.detach().cpu() - y_r.detach()).abs().max())
    e_i, e_c, e_g = rel(i_d.grad, i_r.grad), rel(c_d.grad, c_r.grad), rel(g_d.grad.permute(0, 3, 1, 2), g_r.grad)
    e_s = abs(float(m.aff_scale_const.grad) - float(ref.aff_scale_const.grad)) / max(1.0, abs(float(ref.aff_scale_const.grad)))
    print(f'[nlspn 18 x 228x304] out max|err| {e_out:.2e}; grad rel L2: init {e_i:.2e} conf {e_c:.2e} guidance {e_g:.2e}; aff_scale_const {e_s:.2e}')
    # (the guidance conv runs on the f32 MFMA here and in torch on the host: offsets differ in the last bits, and where a sampling
    #  position sits within that of an integer the OFFSET gradient, which is discontinuous there, takes the other branch)
    assert float((y_r.detach() - init).abs().max()) > 0.3, 'the propagation must move the map'
    assert e_out < 1e-4 and e_i < 1e-3 and e_c < 2e-3 and e_g < 5e-3 and e_s < 2e-3


def test_loss_kernel_against_reference_values(golden):
    """GANLoss('lsgan') / L1_loss / L2_loss (lib/losses/gan_loss.py:6-42, 104-120, 169-206) on the HIP reduction against the values
    and gradients the reference's own functions returned for these inputs (tests/golden/losses.npz, make_golden.py:losses)"""
    from rdfc_gan_amd.gan_loss import GANLoss, L1_loss, L2_loss
    g = golden('losses')
    p = det_tensor(5, 'p', (2, 1, 26, 36)).to(DEV).requires_grad_(True)
    crit = GANLoss('lsgan').to(DEV)
    for real in (True, False):
        v = crit(p, real)
        gr, = torch.autograd.grad(v, p)
        np.testing.assert_allclose(float(v), g[f'gan.lsgan.{int(real)}'], rtol=2e-6)
        np.testing.assert_allclose(gr.cpu().numpy(), g[f'gan.lsgan.{int(real)}.grad'], rtol=1e-5, atol=1e-9)
    for C in (1, 3):
        a = det_tensor(5, f'a{C}', (2, C, 20, 24)).to(DEV).requires_grad_(True)
        b = det_tensor(5, f'b{C}', (2, C, 20, 24)).to(DEV)
        msk = det_tensor(5, 'm', (2, 1, 20, 24), 'rand') > 0.3
        w = (msk / (msk.sum() + 1e-6)).to(DEV)
        for nm, fn in (('l1', L1_loss), ('l2', L2_loss)):
            for wn, ww in (('w', w), ('u', None)):
                v = fn(a, b, weight=ww)
                gr, = torch.autograd.grad(v, a)
                np.testing.assert_allclose(float(v), g[f'{nm}.{C}.{wn}'], rtol=2e-6)
                np.testing.assert_allclose(gr.cpu().numpy(), g[f'{nm}.{C}.{wn}.grad'], rtol=1e-5, atol=1e-9)


def test_build_model_with_the_dict_train_py_assembles():
    """train.py:84-103: model_cfg = dict(type=cfg.model.type, device, distributed, args=cfg, is_train, fixed_*_samples, num_classes,
    out_height, out_width, label_wall, label_floor, label_ceiling) with cfg = upstream's attribute-style config (the YAML of
    config/rdf_cycle_patchgan_config.yaml merged with config.py's defaults) -> build_model(model_cfg).  The object must offer
    the surface train.py / test.py touch (base.py:28-157) and step on a batch with upstream's keys."""
    from rdfc_gan_amd.build_model import build_model
    setup_dtype(torch.bfloat16)

    class Node(dict):
        """attribute-style dict like upstream's cfg (lib/utils/configurator/yacs.py: CfgNode is a dict subclass)"""
        __getattr__ = dict.__getitem__

    def ns(d):
        return Node({k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    yaml_model = dict(type='RDFCGAN',
                      G_A2B=dict(type='RDFGenerator', encoder_rgb='resnet18', encoder_depth='resnet18', pretrained_on_imagenet=True,
                                 semantic_channels_in=3, fuse_depth_in_rgb_decoder='WAdaIN', bn=True, rgb_skip_connection_type='concat',
                                 depth_skip_connection_type='concat', adain_weighting=False,
                                 rgb_channels_encoder=[64, 64, 128, 256, 512, 512], depth_channels_encoder=[64, 64, 128, 256, 512, 512],
                                 rgb_channels_decoder=[256, 128, 64, 64], depth_channels_decoder=[256, 128, 64, 64]),
                      G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                      Normal_G=dict(sampling_ratio=0.4, importance_ratio=0.7, architecture='BN'),
                      D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                      D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'))
    cfg = ns(dict(model=yaml_model, local_rank=0, lr=2e-4, beta1=0.5, beta2=0.999, gan_loss_type='lsgan', lambda_A=10.0, lambda_B=10.0,
                  lambda_L1=100.0, scheduler='linear', epoch=0, decay_epoch=100, n_epochs=200, num_classes=14, out_height=224,
                  out_width=304, label_wall=12, label_floor=5, label_ceiling=3, steps_per_epoch=10, load_from=None))
    model_cfg = dict(type=cfg.model.type, device=torch.device(DEV), distributed=False, args=cfg, is_train=True,
                     fixed_testing_samples=None, fixed_training_samples=None, num_classes=cfg.num_classes, out_height=cfg.out_height,
                     out_width=cfg.out_width, label_wall=cfg.label_wall, label_floor=cfg.label_floor, label_ceiling=cfg.label_ceiling)
    model = build_model(model_cfg)
    for attr in ('set_input', 'optimize_parameters', 'forward_test', '__call__', 'eval', 'train', 'load_from', 'resume_from', 'save_ckpt',
                 'save_samples', 'update_learning_rate', 'generator', 'reduce_loss'):
        assert callable(getattr(model, attr)), attr
    model.train()
    data, nA, _ = step_batch(0, 2, 60, 76)
    data['gt_normal'] = nA
    model.set_input(data)
    losses = model.optimize_parameters()
    assert set(losses) >= {'loss_G', 'loss_D_A', 'loss_D_B', 'loss_G_A2B', 'loss_G_B2A', 'loss_cycle_A', 'loss_cycle_B', 'loss_A2B_L1'} and len(losses) == 13
    assert all(isinstance(v, float) and np.isfinite(v) for v in losses.values())
    model.eval()
    out = model(rgb=data['rgb'], raw_depth=data['raw_depth'], gt_normal=data['gt_normal'])
    assert tuple(out['pred_depth'].shape) == (2, 1, 60, 76)
    with pytest.raises(KeyError):
        build_model(dict(model_cfg, type='NoSuchModel'))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
def test_eval_forward_after_training_steps_and_checkpoint_load(dtype):
    """train.py's flow: optimise (parameters live in FlatAdam's flat buffers, conv weights channels-last, packed operands rebuilt
    in one launch per step), then load a checkpoint's state_dict into the SAME modules, switch to eval() and run the validation
    forward.  The eval outputs must be those of the oracle on the loaded weights -- nothing of the training steps (packed
    operands, BatchNorm bookkeeping, stacked-batch settings) may leak into it."""
    from oracle import nets as on
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    setup_dtype(dtype)
    m = RDFCGAN(torch.device(DEV), False, _cfg(), is_train=True)
    m.train()
    for it in range(2):
        losses = _one_step(m, 2, 60, 76, seed=it)
        assert all(np.isfinite(v) for v in losses.values())
    sd = det_state_dict(m.G_A2B.state_dict(), 3)
    m.G_A2B.load_state_dict(sd)
    Fh.invalidate_packed_weights()
    m.eval()
    ref = on.RDFGenerator()
    ref.load_state_dict(sd)
    ref.eval()
    rgb, raw, normal = gen_inputs('after_train', 2, 60, 76)
    with torch.no_grad():
        exp = ref(rgb, raw, normal)['pred_depth']
    got = m(rgb=rgb, raw_depth=raw, gt_normal=normal)['pred_depth'].float().cpu()
    e = float((got - exp).abs().max())
    print(f'[eval after training] {dtype}: max|err| {e:.2e}')
    assert e < (5e-5 if dtype == torch.float32 else 2e-2), e
    # and training goes on from the loaded weights
    m.train()
    losses = _one_step(m, 2, 60, 76, seed=5)
    assert all(np.isfinite(v) for v in losses.values())
