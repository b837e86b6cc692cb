"""Train-step wrapper for the hot-path nets with upstream's RDFCGAN surface.

Mirrors RDFC-GAN/lib/models/rdfc_gan.py (set_input :66-77, forward :184-207,
backward_G :209-289, backward_D_basic :292-300, optimize_parameters :318-340,
forward_test :168-182) and base.py (:28-157) for G_A2B, G_B2A, disc_A, disc_B.

Outside the hot path: the surface-normal net G_normal and the label
net G_label.  The normals they would produce enter through `normal_provider`
(default: the batch's `gt_normal`, unit-normalised), and the label / normal /
manhattan loss terms are not part of this step.

Differences from upstream that do not change results:
  * upstream's two dead forwards (rdfc_gan.py:197,207 -- recomputed at :228,:234) are
    elided by default.  G_B2A has no running statistics, so its dead forward has no side
    effect at all; G_A2B's dead forward only advances its BatchNorm running statistics a
    second time with the very same batch statistics, which is reproduced exactly by one
    update with momentum 1-(1-m)^2 (`exact_schedule=True` runs upstream's literal schedule).
  * the 23 per-key `all_reduce` + `.item()` calls of Base.reduce_loss (base.py:121-132)
    are one packed all-reduce and one device->host copy per loss group (G, D_A, D_B), posted as soon as the group's
    forward has produced the values (`_LossReadback`): the host gets this step's floats while the GPU still runs the
    discriminators' backward passes and the Adam updates, and starts enqueueing the next step under them.
  * per-step `dist.barrier()` (rdfc_gan.py:337-338) is dropped (the all-reduces order ranks).
"""
import contextlib
import itertools
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import functional as Fh
from . import hip, layers
from .discriminator import build_discriminator
from .gan_loss import GANLoss, L1_loss
from .generator import build_generator
from .init_weights import init_weights
from . import optim
from .optim import FlatAdam, LossScaler, broadcast_module_state, dp_active


def norm_normalize(norm_out):
    """helper.py:404-408"""
    n = torch.sqrt((norm_out ** 2.0).sum(1, keepdim=True)) + 1e-10
    return norm_out / n


_NET_STREAMS = os.environ.get('RDFC_NET_STREAMS', '1') != '0'
# where the discriminators' own step is enqueued (bench.py, bf16, batch 8 at 228x304, three interleaved runs each): '0' behind G's
# backward on the main / B->A streams as upstream orders it: 37.5-37.6 ms per step; '1' behind it on the host but forked before it
# on the B->A stream: 37.0-37.2; 'first' (default) enqueued before it on the depth-branch stream: 36.1-36.4
_D_UNDER_G = {'0': False, '1': True}.get(os.environ.get('RDFC_D_UNDER_G', 'first'), os.environ.get('RDFC_D_UNDER_G', 'first'))


class _LossReadback:
    """base.py:121-132 (mean over ranks, python floats) without draining the queue: every post() packs one group of 0-d
    losses on the stream that produced them, copies it into pinned host memory and records an event; collect() waits for
    those events only.
    Data parallel: the groups are averaged by ONE all-reduce.  A collective issued on a compute stream in mid-step queues
    behind whatever gradient buckets the process group's stream already holds and stalls the compute stream until the weight
    gradients before it have finished -- so it goes out where no bucket of the step has been issued yet: the default schedule
    computes all three groups (G, D_B, D_A) before G's backward and before the discriminators' buckets, and begin(expect=3)
    makes the third post() run the all-reduce and the copy (on the stream of the discriminators' step, which waits for it;
    G's backward does not).  Without `expect` (other schedules) the all-reduce runs at the end of the step, in collect()."""
    SLOTS = 64

    def __init__(self, device):
        self.device = device
        self.host = torch.empty(self.SLOTS, dtype=torch.float32).pin_memory() if device.type == 'cuda' else None
        self.posts, self.used, self.expect, self.flushed = [], 0, None, None

    def begin(self, expect=None):
        self.posts, self.used, self.expect, self.flushed = [], 0, expect, None

    def post(self, stats):
        keys = list(stats)
        packed = torch.stack([stats[k].detach().float().reshape(()) for k in keys])
        if self.host is None or dp_active():
            ev = None
            if packed.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            self.posts.append((keys, packed, ev))
            if dp_active() and self.host is not None and self.expect is not None and len(self.posts) == self.expect:
                keys, packed = self._reduce_dp()
                dst = self.host[:len(keys)]
                dst.copy_(packed, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self.flushed = (keys, dst, ev, packed)
            return
        assert self.used + len(keys) <= self.SLOTS
        dst = self.host[self.used:self.used + len(keys)]
        dst.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.posts.append((keys, dst, ev))
        self.used += len(keys)

    def _reduce_dp(self):
        """mean over the ranks of every group posted so far, on the current stream"""
        for _, _, ev in self.posts:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)              # a group packed on another stream
        keys = [k for ks, _, _ in self.posts for k in ks]
        packed = torch.cat([v for _, v, _ in self.posts]) / dist.get_world_size()
        if not optim.COLLECTIVES_OFF:
            dist.all_reduce(packed)
        return keys, packed

    def collect(self, order):
        got = {}
        if self.flushed is not None:
            keys, vals, ev, _keep = self.flushed
            ev.synchronize()
            got = dict(zip(keys, vals.tolist()))
        elif dp_active():
            keys, packed = self._reduce_dp()
            got = dict(zip(keys, packed.tolist()))
        else:
            for keys, vals, ev in self.posts:
                if ev is not None:
                    ev.synchronize()
                got.update(zip(keys, vals.tolist()))
        self.posts, self.flushed = [], None
        return {k: got[k] for k in order}


def draw_pepper_mask(shape, snr=0.98, p=0.9):
    """host RNG draw of helper.py:384-402 (python `random` + numpy), returned as uint8 (B,1,H,W) or None"""
    if not random.uniform(0, 1) < p:
        return None
    b, c, h, w = shape
    pct = 1 - snr
    # np.random.choice((0, 1, 2), size, p) = searchsorted(cdf(p), random_sample(size), 'right'): the same uniform draws,
    # the same result and the same position of numpy's global stream afterwards, 13x faster than choice()'s generic path
    cdf = np.cumsum(np.array([snr, pct / 2., pct / 2.], dtype=np.float64))
    cdf /= cdf[-1]
    u = np.random.random_sample((b, 1, h, w))
    return (u >= cdf[0]).astype(np.uint8) + (u >= cdf[1]).astype(np.uint8)


class _PepperNoise(torch.autograd.Function):
    """in-place salt/pepper (img[mask==1]=1, img[mask==2]=-1): gradient is cut at the replaced pixels"""

    @staticmethod
    def forward(ctx, img, mask_u8):
        out = hip.apply_pepper(img.contiguous().clone(), mask_u8)
        ctx.save_for_backward(mask_u8)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, = ctx.saved_tensors
        return hip.mask_keep(g, mask), None


def _stack2(a, b):
    """[a | b] along the batch (module inputs: NCHW f32, no gradient needed into `a`/`b` except through autograd's cat)"""
    return _Stack2.apply(a, b)


class _Stack2(torch.autograd.Function):
    """torch.cat((a, b), 0) with device-to-device copies (no elementwise kernel) and a zero-copy backward"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        out[:a.shape[0]].copy_(a)
        out[a.shape[0]:].copy_(b)
        ctx.na = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.na], g[ctx.na:]


class _SplitBatch(torch.autograd.Function):
    """x -> (x[:B], x[B:]) as views; the backward writes the two cotangents into one buffer (copies / memset, no kernels)"""

    @staticmethod
    def forward(ctx, x, B):
        ctx.meta = (tuple(x.shape), x.dtype, x.device, B)
        x = x.contiguous()
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dtype, dev, B = ctx.meta
        g = torch.empty(shape, dtype=dtype, device=dev)
        for part, src in ((g[:B], ga), (g[B:], gb)):
            if src is None:
                hip.zero_(part)
            else:
                part.copy_(src)
        return g, None


def _cfg_get(node, key, default=None):
    if isinstance(node, dict):
        return node.get(key, default)
    return getattr(node, key, default)


class RDFCGAN:
    def __init__(self, device, distributed, args, is_train=True, fixed_training_samples=None, fixed_testing_samples=None,
                 num_classes=14, out_height=224, out_width=304, label_wall=12, label_floor=5, label_ceiling=3,
                 normal_provider=None, exact_schedule=False, compute_dtype=None, use_label_net=False, normal_net=None,
                 device_pepper=False, pepper_seed=0, batch_calls=None, loss_scale=None, dynamic_loss_scale=True):
        self.device, self.distributed, self.args, self.is_train = device, distributed, args, is_train
        # G_label and the label / normal-consistency terms of loss_G (rdfc_gan.py:89,187-203,238-267): off by default -- the
        # benchmarked hot path is the four-network step; on, the step computes upstream's full objective except that the
        # surface normals still come from `normal_provider` / the batch (G_normal is not part of this build)
        self.use_label_net = bool(use_label_net)
        # addPepperNoise's mask (helper.py:384-402) drawn on the device instead of from numpy's global stream + a host copy: same
        # distribution, different random numbers -- off by default so that runs reproduce upstream's draws
        self.device_pepper, self._pepper_seed, self._pepper_calls = bool(device_pepper), int(pepper_seed), 0
        # G_normal (rdfc_gan.py:88): an NNET-like module (rdfc_gan_amd.normal_net.NNET around the caller's encoder).  With it the
        # normals are predicted -- fake_normal = norm_normalize(G_normal(img)[:, 0:3]) (:190-191, 205-206, 233-234) -- the normal
        # terms back-propagate into it and it gets upstream's AdamW (:141-144)
        self.G_normal = normal_net
        self.num_classes, self.label_wall, self.label_floor, self.label_ceiling = num_classes, label_wall, label_floor, label_ceiling
        self.local_rank = _cfg_get(args, 'local_rank', 0)
        self.out_height, self.out_width = out_height, out_width
        self.normal_provider = normal_provider
        self.exact_schedule = exact_schedule
        # Batched schedule (four-network hot path only): each network's two forward / backward passes per step run as ONE pass
        # over the two inputs stacked along the batch -- G_A2B([real_A | fake_A]), G_B2A([real_B | fake_B]), D([real | fake]) --
        # with BatchNorm statistics per half (layers.BN_GROUP_REPEATS): half as many, twice as large launches for the same
        # per-sample arithmetic.  G_A2B never reads its `rgb` argument (rdf_generator.py:282), so its second pass does not
        # depend on fake_A unless the surface normals are computed from it: None = on when the normals come from the batch
        # (gt_normal); True = also with a `normal_provider`, which is then called with img=None for the fake_A tags.
        self.batch_calls = batch_calls
        if compute_dtype is not None:
            Fh.set_compute_dtype(compute_dtype)
        # Loss scaling (f16 mode only -- the per-pixel loss gradients, ~1e-6, sit below half precision's normal range; f32 / bf16
        # keep 1): every backward starts from the cotangent `scale` instead of 1 and the optimizers divide it out again inside
        # the Adam kernel.  The scale lives on the device (optim.LossScaler): the optimizers check their gradients for inf / NaN
        # there, skip an overflowing step and -- dynamic_loss_scale, the default -- halve the scale after an overflow / double it
        # after 2 000 clean steps.  loss_scale: the initial (or, dynamic_loss_scale=False, the fixed) value, default 1024.
        if loss_scale is None:
            loss_scale = 1024.0 if Fh.compute_dtype() == torch.float16 else 1.0
        self._init_loss_scale, self._dynamic_loss_scale = float(loss_scale), bool(dynamic_loss_scale)
        self.scaler = None
        self.init_model()
        if self.is_train:
            self.init_optimizer()
            self.init_lr_scheduler()
            self.criterionGAN = GANLoss(_cfg_get(args, 'gan_loss_type', 'lsgan')).to(device)
            self.criterionCycle = L1_loss
            self.criterionL1 = L1_loss
            self.fixed_training_samples, self.fixed_testing_samples = fixed_training_samples, fixed_testing_samples

    # ---- construction ------------------------------------------------------------------------
    def init_model(self):
        m = _cfg_get(self.args, 'model')
        self.G_A2B = build_generator(_cfg_get(m, 'G_A2B')).to(self.device)
        self.G_B2A = build_generator(_cfg_get(m, 'G_B2A')).to(self.device)
        self.disc_A = build_discriminator(_cfg_get(m, 'D_A')).to(self.device)
        self.disc_B = build_discriminator(_cfg_get(m, 'D_B')).to(self.device)
        for net in (self.G_A2B, self.G_B2A, self.disc_A, self.disc_B):
            init_weights(net, verbose=False)
        self.models = dict(G_A2B=self.G_A2B, G_B2A=self.G_B2A, disc_A=self.disc_A, disc_B=self.disc_B)
        if self.use_label_net:
            from .label_net import PSPNet
            self.G_label = PSPNet(self.num_classes).to(self.device)              # rdfc_gan.py:89
            self.models['G_label'] = self.G_label
        if self.G_normal is not None:
            self.G_normal = self.G_normal.to(self.device)
            self.models['G_normal'] = self.G_normal
        if self.distributed:
            # DDP-constructor semantics without DDP: same start on every rank; gradients are
            # exchanged by FlatAdam.all_reduce_grads_async (one bucket per optimizer)
            broadcast_module_state(self.models.values())

    def init_optimizer(self):
        a = self.args
        lr, betas = _cfg_get(a, 'lr', 2e-3), (_cfg_get(a, 'beta1', 0.5), _cfg_get(a, 'beta2', 0.999))
        self.optimizer_G = FlatAdam(itertools.chain(self.G_A2B.parameters(), self.G_B2A.parameters()), lr=lr, betas=betas)
        self.optimizer_D = FlatAdam(itertools.chain(self.disc_A.parameters(), self.disc_B.parameters()), lr=lr, betas=betas)
        self.optimizers = dict(G=self.optimizer_G, D=self.optimizer_D)
        if self._init_loss_scale != 1.0:
            self.scaler = LossScaler(self.device, self._init_loss_scale, dynamic=self._dynamic_loss_scale)
            for o in self.optimizers.values():
                self.scaler.attach(o)
        # Data-parallel exchange in buckets, in the order in which the gradients become final (upstream: DDP's 25 MB
        # reverse-order buckets, rdfc_gan.py:102-119), each started from INSIDE the backward by a marker (functional.backward_marker):
        #   0  G_B2A (31 MB): its backward completes first (marker on its input);
        #   1  G_A2B's decoder side -- decoders, W-AdaIN fusion layers, heads, NLSPN (~27 MB): final when the backward of both
        #      branches has passed the en6 outputs (markers in RDFGenerator.forward);
        #   2  G_A2B's deep encoder stages en5 + en6 of both branches (~86 MB, two thirds of the network's parameters): final when
        #      the backward has passed the inputs of en5;
        #   3  the rest (en1 .. en4, ~24 MB): at the end of the backward.
        # Round 3 cut G_A2B into three chunks in PARAMETER order and started all of them when the backward had ended.
        oG, oD = self.optimizer_G, self.optimizer_D
        a0, a1 = oG.param_range(list(self.G_A2B.parameters()))
        b0, b1 = oG.param_range(list(self.G_B2A.parameters()))
        G = self.G_A2B
        if hasattr(G, 'rgb_branch_encoder_decoder'):

            def rng(mods):
                return [oG.param_range(list(m.parameters())) for m in mods if next(m.parameters(), None) is not None]
            R, Dn = G.rgb_branch_encoder_decoder, G.depth_branch_encoder_decoder
            deep = rng([R.en5, R.en6, Dn.en5, Dn.en6])
            shallow = rng([G.rgb_branch_en1, R.en2, R.en3, R.en4, G.depth_branch_en1_rgb, G.depth_branch_en1_depth, Dn.en2, Dn.en3, Dn.en4])
            dec, at = [], a0                                  # everything of G_A2B that is neither: the decoder side
            for lo, hi in sorted(deep + shallow):
                if lo > at:
                    dec.append((at, lo))
                at = hi
            if at < a1:
                dec.append((at, a1))
            oG.set_buckets([[(b0, b1)], dec, deep, shallow])
            self._G_bucket = dict(G_B2A=0, dec=1, deep=2, rest=3)
        else:                                                 # any other generator class: one bucket behind G_B2A's
            oG.set_buckets([(b0, b1), (a0, a1)])
            self._G_bucket = dict(G_B2A=0, rest=1)
        oD.set_buckets([oD.param_range(list(self.disc_A.parameters())), oD.param_range(list(self.disc_B.parameters()))])
        if self.use_label_net:
            self.optimizer_Label = FlatAdam(self.G_label.parameters(), lr=lr, betas=betas)   # rdfc_gan.py:145-147
            self.optimizers['label'] = self.optimizer_Label
            if self.scaler is not None:
                self.scaler.attach(self.optimizer_Label)
        if self.G_normal is not None:                                            # torch.optim.AdamW, lr / 2 (:141-144)
            self.optimizer_Normal = FlatAdam(self.G_normal.parameters(), lr=lr * 0.5, betas=betas,
                                             weight_decay=_cfg_get(a, 'weight_decay', 0.0))
            self.optimizers['Nor'] = self.optimizer_Normal
            if self.scaler is not None:
                self.scaler.attach(self.optimizer_Normal)

    def init_lr_scheduler(self):
        a = self.args
        sched = str(_cfg_get(a, 'scheduler', 'linear')).lower()
        if sched == 'linear':
            e0, dec, n = _cfg_get(a, 'epoch', 0), _cfg_get(a, 'decay_epoch', 100), _cfg_get(a, 'n_epochs', 150)
            make = lambda opt: torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda ep: 1.0 - max(0, ep + e0 - dec) / (n - dec))
        elif sched == 'step':
            make = lambda opt: torch.optim.lr_scheduler.MultiStepLR(opt, milestones=_cfg_get(a, 'lr_decay_epochs'),
                                                                    gamma=_cfg_get(a, 'lr_decay_rate'))
        elif sched == 'cosine':
            make = lambda opt: torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=_cfg_get(a, 'n_epochs', 150), eta_min=0)
        else:
            raise NotImplementedError
        self.lr_scheduler_G, self.lr_scheduler_D = make(self.optimizer_G), make(self.optimizer_D)
        self.lr_schedulers = dict(G=self.lr_scheduler_G, D=self.lr_scheduler_D)
        if self.use_label_net:
            self.lr_scheduler_Label = make(self.optimizer_Label)
            self.lr_schedulers['label'] = self.lr_scheduler_Label
        if self.G_normal is not None:
            self.lr_scheduler_Nor = make(self.optimizer_Normal)
            self.lr_schedulers['Nor'] = self.lr_scheduler_Nor

    # ---- data ------------------------------------------------------------------------------------
    def set_input(self, data):
        if getattr(self, '_graph_mode', False):
            return self._set_input_static(data)
        dev = self.device
        self.real_A, self.real_B = data['rgb'].to(dev), data['gt_depth'].to(dev)
        self.aux_A = data['raw_depth'].to(dev)
        self.gt_normal = data['gt_normal'].to(dev) if 'gt_normal' in data else None
        self.mask = data['depth_masks'].to(dev) if 'depth_masks' in data else torch.ones_like(self.real_B)
        self.image_loss_weight = self.mask / (self.mask.sum() + 1e-6)
        if self.use_label_net:
            self.gt_label = data['labels'].to(dev)                               # rdfc_gan.py set_input: labels, normal_masks
        if self.use_label_net or self.G_normal is not None:
            self.norm_mask = data['normal_masks'].to(dev) if 'normal_masks' in data else None
        self._batch = data
        self._unit_gt = None

    def _set_input_static(self, data):
        """graph mode: the captured step reads its batch from fixed device buffers; a new batch is copied into them"""
        dev = self.device
        st = self.__dict__.setdefault('_static_in', {})
        src = dict(real_A=data['rgb'], real_B=data['gt_depth'], aux_A=data['raw_depth'])
        if 'gt_normal' in data:
            src['gt_normal'] = data['gt_normal']
        src['mask'] = data['depth_masks'] if 'depth_masks' in data else torch.ones_like(data['gt_depth'])
        for k, v in src.items():
            if k not in st or st[k].shape != v.shape:
                st[k] = v.to(dev).clone()
                self._graphs = None                      # shapes changed: capture again
            else:
                st[k].copy_(v, non_blocking=True)
        self.real_A, self.real_B, self.aux_A, self.mask = st['real_A'], st['real_B'], st['aux_A'], st['mask']
        self.gt_normal = st.get('gt_normal')
        if 'image_loss_weight' not in st or st['image_loss_weight'].shape != self.mask.shape:
            st['image_loss_weight'] = torch.empty(self.mask.shape, dtype=torch.float32, device=dev)
        st['image_loss_weight'].copy_(self.mask / (self.mask.sum() + 1e-6))
        self.image_loss_weight = st['image_loss_weight']
        self._batch = data
        self._unit_gt = None

    def _normal(self, tag, img):
        if self.G_normal is not None:
            return norm_normalize(self.G_normal(img)[:, 0:3])
        if self.normal_provider is not None:
            v = self.normal_provider(tag, img)
            return norm_normalize(v[:, 0:3]) if v.requires_grad or not v.is_cuda else hip.unit_normals(v)
        if self.gt_normal is None:
            raise RuntimeError('no normal_provider and no gt_normal in the batch: the hot path takes surface normals as input')
        if not self.gt_normal.is_cuda:
            return norm_normalize(self.gt_normal[:, 0:3])
        hit = self.__dict__.get('_unit_gt')        # same batch, same normals: normalised once per set_input
        if hit is None or hit[0] is not self.gt_normal:
            hit = self._unit_gt = (self.gt_normal, hip.unit_normals(self.gt_normal))
        return hit[1]

    # ---- inference (forward_test :168-182) -----------------------------------------------------------
    @torch.no_grad()
    def forward_test(self, **kwargs):
        rgb, raw = kwargs['rgb'].to(self.device), kwargs['raw_depth'].to(self.device)
        if 'gt_normal' in kwargs:
            self.gt_normal = kwargs['gt_normal'].to(self.device)
        pred_normal = self._normal('real_A', rgb)
        ret = self.G_A2B(rgb, raw, pred_normal)
        out = dict(pred_depth=ret['pred_depth'], pred_normal=pred_normal)
        if self.use_label_net:
            out['pred_label'] = self.G_label(rgb)
        return out

    def __call__(self, **kwargs):
        return self.forward_test(**kwargs)

    def generator(self):
        return self.G_A2B

    # ---- one optimisation step -----------------------------------------------------------------------
    def set_requires_grad(self, models, requires_grad=False):
        for m in (models if isinstance(models, list) else [models]):
            if m is not None:
                for p in m.parameters():
                    p.requires_grad = requires_grad

    def _batched(self):
        if self.exact_schedule or self.use_label_net or self.G_normal is not None or self.batch_calls is False:
            return False
        return True if self.batch_calls else self.normal_provider is None

    def _forward_batched(self):
        """forward() of the batched schedule: fake_B | rec_B from one G_A2B pass, fake_A | rec_A from one G_B2A pass"""
        B = self.real_B.shape[0]
        # The pepper mask of fake_B (helper.py:384-402, a host RNG draw of B x H x W doubles: 1.5 ms) does not depend on fake_B: draw and
        # upload it BEFORE G_A2B's forward is enqueued, not between that forward and the cycle pass where the host stalls with nothing
        # queued behind the generator.  Nothing between here and upstream's draw point consumes python / numpy random numbers, so the
        # stream positions -- and the image pools' draws after it -- are unchanged.
        self._pepper_pre = None
        if not getattr(self, '_graph_mode', False) and not self.device_pepper:
            mask = draw_pepper_mask((B, 1) + tuple(self.real_B.shape[2:]))
            self._pepper_pre = (None if mask is None else self._mask_to_device(mask),)
        self.fake_normal_realA = self._normal('real_A', self.real_A)
        self.fake_normal_fakeA = self._normal('fake_A', None)
        normal_again = self._normal('fake_A_again', None)
        # group 0 = G_A2B(real_A, ...) of forward() (:193); group 1 = G_A2B(fake_A, ...): the dead forward (:207) and the
        # recomputation in backward_G (:234) see the same batch statistics -> two running-statistics updates
        # (under data parallelism G_A2B's backward starts its own gradient buckets: markers at the decoder / encoder boundary and
        #  in front of the deep encoder stages of both branches, init_optimizer)
        self._G_marks = dict(dec=0, deep=0)
        self._G_mark_events = {}
        self.G_A2B.__dict__['bwd_marks'] = self._on_G_mark if (dp_active() and self.is_train and not getattr(self, '_graph_mode', False)) else None
        try:
            with layers.bn_groups((1, 2)):
                out = self.G_A2B(None, _stack2(self.aux_A, self.aux_A), _stack2(self.fake_normal_realA, normal_again))['pred_depth']
        finally:
            self.G_A2B.__dict__['bwd_marks'] = None
        fake_B, self.rec_B = _SplitBatch.apply(out, B)
        self.fake_B = self._pepper(fake_B)
        # fake_B feeds G_B2A, D_A and the L1 term; fake_A feeds D_B and its L1 term: their cotangents are summed by one HIP pass
        fB_cyc, self._fB_D, self._fB_L1 = Fh.fanout_flat(self.fake_B, 3)
        # G_B2A is InstanceNorm only: nothing in it couples the samples of a batch
        # (the marker's backward runs when G_B2A's backward has been enqueued: its gradient bucket starts its all-reduce then)
        # (Round 3 tried the two passes un-stacked, G_B2A(real_B) enqueued first on the B->A stream beside G_A2B's forward:
        #  210.9 / 211.5 pairs/s against 218.4 / 213.6 stacked -- more overlap does not buy time here)
        out = self.G_B2A(Fh.backward_marker(_stack2(self.real_B, fB_cyc), self._start_G_B2A_bucket))
        self.fake_A, self.rec_A = _SplitBatch.apply(out, B)
        self._fA_D, self._fA_L1 = Fh.fanout_flat(self.fake_A, 2)

    def _on_G_mark(self, name):
        """called from inside G_A2B's backward (RDFGenerator.forward's markers), once per branch: when both branches have passed the
        point, the bucket whose gradients are final by then starts its all-reduce (the weight-gradient stream, from which it is
        issued, has the layers' weight gradients in front of it; the other branch's stream is waited for: helpers=True)"""
        self._G_marks[name] += 1
        # every marker leaves an event on the stream it fired on: the bucket follows BOTH branches' streams, not only the one whose
        # marker happens to fire last (the gradients autograd accumulates behind the other branch's last node are on that stream)
        if self.device.type == 'cuda':           # (the gloo host-logic tests drive the markers on CPU tensors: nothing to order there)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._G_mark_events.setdefault(name, []).append(ev)
        if self._G_marks[name] == 2 and name in self._G_bucket:
            self.optimizer_G.all_reduce_bucket_async(self._G_bucket[name], after=self._G_mark_events.pop(name, ()))

    def _start_G_B2A_bucket(self):
        if self.is_train and not getattr(self, '_graph_mode', False):
            # (G_B2A's backward runs on the current stream and the weight-gradient stream only: no reason to hold the
            #  weight gradients until the depth-branch stream -- busy with the discriminators' step -- has caught up)
            self.optimizer_G.all_reduce_bucket_async(0, helpers=not self._batched())

    def _pepper(self, fake_B):
        if getattr(self, '_graph_mode', False):
            # host RNG draw happened before the replay (_draw_pepper_static); an all-zero mask is the "no noise" draw
            return _PepperNoise.apply(fake_B, self._pepper_static)
        if self.device_pepper:
            self._pepper_calls += 1
            if random.uniform(0, 1) < 0.9:                                       # the per-call coin of helper.py:386 stays on the host
                b, _, h, w = fake_B.shape
                mask = hip.draw_pepper_mask(b, h, w, self._pepper_seed + 7919 * self.local_rank, self._pepper_calls, 0.98, self.device)
                return _PepperNoise.apply(fake_B, mask)
            return fake_B
        pre, self._pepper_pre = getattr(self, '_pepper_pre', None), None
        if pre is not None:                  # drawn at the top of _forward_batched
            return fake_B if pre[0] is None else _PepperNoise.apply(fake_B, pre[0])
        mask = draw_pepper_mask(fake_B.shape)
        return fake_B if mask is None else _PepperNoise.apply(fake_B, self._mask_to_device(mask))

    def forward(self):
        if self._batched():
            return self._forward_batched()
        # The B->A half (G_B2A(real_B)) shares nothing with the A->B half in this method when only the four hot-path networks
        # run: it forks HERE, not behind everything the main stream has enqueued by then.  (G_label / G_normal are used by
        # both halves -- running statistics, packed weights -- so with them the fork stays behind the A->B half.)
        fork = None
        if self._net_streams_on() and not self.use_label_net and self.G_normal is None:
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(self.device))
        self.fake_normal_realA = self._normal('real_A', self.real_A)
        if self.use_label_net:
            # :187 (a fifth stream for this pass -- it depends on nothing but the image -- measured slower: 71-73 vs 67.7 ms per step)
            self.fake_label_realA_list = self.G_label(self.real_A)
        self.fake_B = self._pepper(self.G_A2B(self.real_A, self.aux_A, self.fake_normal_realA)['pred_depth'])
        if self.exact_schedule:
            with torch.no_grad():
                self.G_B2A(self.fake_B)                                          # dead forward (:197)
        with self._net_stream_ctx(fork):                                         # B->A does not depend on A->B: own stream
            self.fake_A = self.G_B2A(self.real_B)
            self.fake_normal_fakeA = self._normal('fake_A', self.fake_A)
            if self.use_label_net:
                self.fake_label_fakeA_list = self.G_label(self.fake_A)           # :201 (its loss reaches G_B2A through fake_A)
        if self.exact_schedule:
            with torch.no_grad():
                self.G_A2B(self.fake_A, self.aux_A, self.fake_normal_fakeA)      # dead forward (:207)

    # ---- two independent halves of the G step on two HIP streams (RDFC_NET_STREAMS=0: one stream) --------------------
    # fake_B = G_A2B(real_A) -> D_A(fake_B), L1, G_B2A(fake_B) cycle      |   fake_A = G_B2A(real_B) -> D_B(fake_A), L1,
    # G_A2B(fake_A) cycle.  The halves share parameters but no activations; each network's two uses stay ordered (packed
    # weights, BatchNorm running statistics) because the streams are joined between forward() and backward_G().
    def _net_streams_on(self):
        return _NET_STREAMS and Fh.multi_stream() and self.device.type == 'cuda' and not self.exact_schedule

    @contextlib.contextmanager
    def _net_stream_ctx(self, fork_event=None):
        if not self._net_streams_on():
            yield
            return
        main, side = torch.cuda.current_stream(self.device), Fh.net_stream(self.device)
        if fork_event is not None:
            side.wait_event(fork_event)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            yield

    def _join_net_stream(self, tensors=()):
        if self._net_streams_on():
            main, side = torch.cuda.current_stream(self.device), Fh.net_stream(self.device)
            main.wait_stream(side)
            for t in tensors:
                if t is not None:
                    t.record_stream(main)        # allocated on the side stream, read on the main one

    def _backward_G_batched(self):
        a = self.args
        lam_A, lam_B, lam_1 = _cfg_get(a, 'lambda_A', 100.0), _cfg_get(a, 'lambda_B', 100.0), _cfg_get(a, 'lambda_L1', 100.0)
        # D_B's G-loss pass on the B->A stream -- unless a process group exists: the runtime has four hardware queues, RCCL's
        # stream needs one, and a compute stream that shares a queue with it stalls behind the collectives' waits
        # (bench.py: 36.45 vs 36.85 ms per step with / without the fourth compute stream; one-rank RCCL 38.6 vs 37.3)
        gloss_net = not (dist.is_available() and dist.is_initialized())
        gloss_net = gloss_net and self._net_streams_on()
        if gloss_net:
            with self._net_stream_ctx():
                gan_B2A = self.criterionGAN(self.disc_B(self._fA_D), True)
        else:
            gan_B2A = self.criterionGAN(self.disc_B(self._fA_D), True)
        gan_A2B = self.criterionGAN(self.disc_A(self._fB_D), True)
        l1_B2A = self.criterionL1(self._fA_L1, self.real_A, weight=self.image_loss_weight)
        cyc_B2A = self.criterionCycle(self.rec_B, self.real_B, weight=self.image_loss_weight)
        l1_A2B = self.criterionL1(self._fB_L1, self.real_B, weight=self.image_loss_weight)
        cyc_A2B = self.criterionCycle(self.rec_A, self.real_A)
        if gloss_net:
            self._join_net_stream((gan_B2A,))
        # loss_G = loss_G_A2B + loss_cycle_A2B + loss_cycle_B2A + loss_G_B2A + loss_A2B_L1 + loss_B2A_L1 (:252-254), the
        # lambda factors folded into one f32 product per term ((x * lambda_L1) * 5 upstream: at most one rounding apart)
        loss_G, terms = Fh.combine_losses([(gan_A2B, 1.0), (cyc_A2B, lam_A), (cyc_B2A, lam_B), (gan_B2A, 1.0),
                                           (l1_A2B, lam_1 * 5), (l1_B2A, lam_1 * 3)])
        stats = dict(loss_G=loss_G, loss_G_A2B=terms[0], loss_G_B2A=terms[3], loss_B2A_L1=terms[5],
                     loss_A2B_L1=terms[5],             # upstream logs B2A under this key (rdfc_gan.py:276)
                     loss_cycle_A=terms[1], loss_cycle_B=terms[2])
        self._post_losses(stats)
        # the discriminators' own step needs nothing of G's backward: it forks HERE (see _enqueue_step)
        self._fork_D = None
        if _D_UNDER_G and Fh.multi_stream() and self.device.type == 'cuda':
            self._fork_D = torch.cuda.Event()
            self._fork_D.record(torch.cuda.current_stream(self.device))
            if _D_UNDER_G == 'first' and self.__dict__.get('_D_hook') is not None:
                self._D_hook()                       # enqueue the discriminators' step before G's backward
        self._backward(loss_G)
        return stats

    def backward_G(self):
        self._fork_D = None
        if self._batched():
            return self._backward_G_batched()
        a = self.args
        lam_A, lam_B, lam_1 = _cfg_get(a, 'lambda_A', 100.0), _cfg_get(a, 'lambda_B', 100.0), _cfg_get(a, 'lambda_L1', 100.0)
        self._join_net_stream((self.fake_A, self.fake_normal_fakeA))             # forward()'s B->A half
        with self._net_stream_ctx():
            loss_G_B2A = self.criterionGAN(self.disc_B(self.fake_A), True)
            loss_B2A_L1 = self.criterionL1(self.fake_A, self.real_A, weight=self.image_loss_weight) * lam_1 * 3
            fakeA_normal = self._normal('fake_A_again', self.fake_A)
            layers.BN_STAT_REPEAT = 1 if self.exact_schedule else 2              # stands in for the elided dead forward
            try:
                rec_B = self.G_A2B(self.fake_A, self.aux_A, fakeA_normal)['pred_depth']
            finally:
                layers.BN_STAT_REPEAT = 1
            loss_cycle_B2A = self.criterionCycle(rec_B, self.real_B, weight=self.image_loss_weight) * lam_B
            extra_B = self._label_and_normal_terms('B2A', getattr(self, 'fake_label_fakeA_list', None), self.fake_normal_fakeA, lam_1)
        loss_G_A2B = self.criterionGAN(self.disc_A(self.fake_B), True)
        loss_A2B_L1 = self.criterionL1(self.fake_B, self.real_B, weight=self.image_loss_weight) * lam_1 * 5
        rec_A = self.G_B2A(self.fake_B)
        loss_cycle_A2B = self.criterionCycle(rec_A, self.real_A) * lam_A
        extra_A = self._label_and_normal_terms('A2B', getattr(self, 'fake_label_realA_list', None), self.fake_normal_realA, lam_1)
        self._join_net_stream((loss_G_B2A, loss_B2A_L1, loss_cycle_B2A) + tuple(extra_B.values()))
        loss_G = loss_G_A2B + loss_cycle_A2B + loss_cycle_B2A + loss_G_B2A + loss_A2B_L1 + loss_B2A_L1
        for v in list(extra_A.values()) + list(extra_B.values()):               # :252-267
            loss_G = loss_G + v
        stats = dict(loss_G=loss_G, loss_G_A2B=loss_G_A2B, loss_G_B2A=loss_G_B2A, loss_B2A_L1=loss_B2A_L1,
                     loss_A2B_L1=loss_B2A_L1,          # upstream logs B2A under this key (rdfc_gan.py:276)
                     loss_cycle_A=loss_cycle_A2B, loss_cycle_B=loss_cycle_B2A)
        stats.update(extra_A)
        stats.update(extra_B)
        self._post_losses(stats)
        self._backward(loss_G)
        return stats

    def _label_and_normal_terms(self, tag, label_list, normal, lam_1):
        """rdfc_gan.py:238-250 for one direction: cross entropy of the label net's two heads (aux x 0.4), nor_loss x 2 and the
        three manhattan terms, all x lambda_L1.  Without G_label there are no label / manhattan terms; without G_label and
        G_normal nothing is added (the four-network hot path)."""
        if not self.use_label_net and self.G_normal is None:
            return {}
        out = {}
        logits = None
        if self.use_label_net:
            ce = Fh.cross_entropy_nchw
            out[f'loss_label_{tag}'] = (ce(label_list[0], self.gt_label) + ce(label_list[1], self.gt_label) * 0.4) * lam_1
            logits = label_list[0]
        labels3 = (self.label_wall, self.label_floor, self.label_ceiling)
        if normal.requires_grad:       # predicted normals: the terms back-propagate into G_normal
            v = Fh.normal_losses(normal, self.gt_normal, self.norm_mask, logits, *labels3)
        else:
            v = hip.normal_losses(normal, self.gt_normal, self.norm_mask, logits, *labels3)
        out[f'loss_normal_{tag}'] = v[0] * (lam_1 * 2)
        if self.use_label_net:
            out.update({f'floor_loss_{tag}': v[1] * lam_1, f'wall_loss_{tag}': v[2] * lam_1, f'ceiling_loss_{tag}': v[3] * lam_1})
        return out

    def _post_losses(self, stats):
        """hand a finished group of loss values to the step's read-back (no-op outside optimize_parameters(sync_losses=True))"""
        rb = self.__dict__.get('_rb_active')
        if rb is not None:
            rb.post(stats)

    def backward_D_basic(self, model_D, real, fake, names=None):
        """names: the three logging keys (loss, real, fake) under which the values go to the step's loss read-back before the
        backward pass is enqueued"""
        post = (lambda d, r, f: self._post_losses({names[0]: d, names[1]: r, names[2]: f})) if names else (lambda *a: None)
        if self._batched():
            # D(real) then D(fake.detach()) as one pass over [real | fake]: BatchNorm statistics (and running-statistics
            # updates, in upstream's order) per half
            with layers.bn_groups((1, 1)):
                pred = model_D(_stack2(real, fake.detach()))
            pred_real, pred_fake = _SplitBatch.apply(pred, real.shape[0])
            loss_D_real = self.criterionGAN(pred_real, True)
            loss_D_fake = self.criterionGAN(pred_fake, False)
            loss_D, _ = Fh.combine_losses([(loss_D_real, 1.0), (loss_D_fake, 1.0)], post=0.5)    # (real + fake) * 0.5
            post(loss_D, loss_D_real, loss_D_fake)
            self._backward(loss_D)
            return loss_D_real, loss_D_fake, loss_D
        loss_D_real = self.criterionGAN(model_D(real), True)
        loss_D_fake = self.criterionGAN(model_D(fake.detach()), False)
        loss_D = (loss_D_real + loss_D_fake) * 0.5
        post(loss_D, loss_D_real, loss_D_fake)
        self._backward(loss_D)
        return loss_D_real, loss_D_fake, loss_D

    @property
    def loss_scale(self):
        """current loss scale (reads the device value back: reporting only)"""
        return self.scaler.scale() if self.scaler is not None else 1.0

    def _backward(self, loss):
        if self.scaler is None:
            loss.backward()
        else:
            loss.backward(gradient=self.scaler.seed)

    def backward_D_A2B(self):
        r, f, d = self.backward_D_basic(self.disc_A, self.real_B, self.fake_B, names=('loss_D_A', 'loss_D_A_real', 'loss_D_A_fake'))
        return dict(loss_D_A=d, loss_D_A_real=r, loss_D_A_fake=f)

    def backward_D_B2A(self):
        r, f, d = self.backward_D_basic(self.disc_B, self.real_A, self.fake_A, names=('loss_D_B', 'loss_D_B_real', 'loss_D_B_fake'))
        return dict(loss_D_B=d, loss_D_B_real=r, loss_D_B_fake=f)

    # ---- HIP-graph mode: the step is captured once as three graphs (G step | D step | Adam) and replayed; the
    #      data-parallel all-reduces stay eager between them, the host only draws the pepper mask and copies the batch in
    def enable_graph(self, enabled=True):
        """everything is captured on one stream.  (Round 2 also captured the eager step's fork / join structure over the helper
        streams: with any ONE helper stream captured the replay works but is no faster than eager launches -- 46-56 ms per step
        against 40-44, hipGraphLaunch costs the host about as much per node as a launch -- and with all of them
        hipStreamEndCapture segfaults on ROCm 7.2; that mode was removed in round 5.  Graph mode is not the benchmark's default.)"""
        if enabled and (self.use_label_net or self.G_normal is not None):
            raise NotImplementedError('graph mode captures the four-network step only')
        self._graph_mode = bool(enabled)
        self._graphs = None
        Fh.set_multi_stream(not enabled)
        if enabled:
            for o in self.optimizers.values():
                o.make_capturable()

    def _mask_to_device(self, mask):
        """host mask -> device through a pinned staging buffer: a pageable-memory copy makes the host wait for every kernel
        queued before it, i.e. ends the host's run-ahead once per step"""
        if self.device.type != 'cuda':
            return torch.from_numpy(mask).to(self.device)
        ring = self.__dict__.setdefault('_pin_ring', [])
        if not ring or ring[0][0].shape != mask.shape:
            ring[:] = [[torch.empty(mask.shape, dtype=torch.uint8).pin_memory(), None] for _ in range(4)]
            self._pin_next = 0
        slot = ring[self._pin_next % len(ring)]
        self._pin_next += 1
        if slot[1] is not None:
            slot[1].synchronize()            # the copy that last used this staging buffer (4 steps ago)
        slot[0].numpy()[...] = mask
        dev = slot[0].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return dev

    def _draw_pepper_static(self):
        shape = self.real_B.shape
        buf = self.__dict__.get('_pepper_static')
        if buf is None or buf.shape != (shape[0], 1, shape[2], shape[3]):
            buf = self._pepper_static = torch.zeros(shape[0], 1, shape[2], shape[3], dtype=torch.uint8, device=self.device)
            self._graphs = None
        mask = draw_pepper_mask(shape)
        if mask is None:
            buf.zero_()
        else:
            buf.copy_(self._mask_to_device(mask))

    def _seg_G(self):
        self.forward()
        self.set_requires_grad([self.disc_A, self.disc_B], False)
        self.optimizer_G.zero_grad()
        stats = self.backward_G()
        Fh.join_side_stream()               # a captured segment ends with every forked stream joined
        return stats

    def _seg_D(self):
        self.set_requires_grad([self.disc_A, self.disc_B], True)
        self.optimizer_D.zero_grad()
        with self._net_stream_ctx():                  # as in the eager step: D_B's step on the second stream
            stats_DB = self.backward_D_B2A()
        stats = self.backward_D_A2B()
        self._join_net_stream(tuple(stats_DB.values()))
        stats.update(stats_DB)
        Fh.join_side_stream()
        return stats

    def _seg_adam(self):
        self.optimizer_G.step()
        self.optimizer_D.step()
        if self.scaler is not None:
            self.scaler.update()

    def _bn_modules(self):
        return [m for net in self.models.values() for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]

    def _capture(self):
        # Autograd nodes of an earlier eager step (AccumulateGrad is bound to the stream it was created on) must be gone
        # before the capture stream builds its own: drop every tensor that still holds that step's graph.
        import gc
        self.fake_A = self.fake_B = self.fake_normal_fakeA = None
        self.rec_A = self.rec_B = self._fA_D = self._fA_L1 = self._fB_D = self._fB_L1 = None
        self._D_hook = self._D_stats = self._fork_D = None
        Fh.invalidate_packed_weights()
        gc.collect()
        bns = self._bn_modules()
        before = [m.__dict__.get('_rdfc_pending_batches', 0) for m in bns]
        for o in self.optimizers.values():
            o.wait_grads()
            o.host_counts = False
        torch.cuda.synchronize()
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread polls events while we capture; only this thread's calls belong to the graph
        # one capture stream for the three graphs, given the deterministic weight-gradient kernels' scratch slab first (nothing can be
        # allocated inside a capture: without it every captured weight gradient would fall back to the float-atomics kernels)
        cs = self.__dict__.get('_capture_stream')
        if cs is None:
            cs = self._capture_stream = torch.cuda.Stream(self.device)
        hip.wgrad_ring_reserve(cs)
        mode = dict(capture_error_mode='thread_local', stream=cs)
        try:
            with torch.cuda.graph(g1, **mode):
                stats = self._seg_G()
            with torch.cuda.graph(g2, pool=g1.pool(), **mode):
                stats.update(self._seg_D())
            with torch.cuda.graph(g3, pool=g1.pool(), **mode):
                self._seg_adam()
        finally:
            for o in self.optimizers.values():
                o.host_counts = True
        # capture ran the host code once without executing anything: undo its host-side bookkeeping and remember the
        # per-step BatchNorm batch counts so that every replay can re-apply them
        self._bn_counts = []
        for m, b in zip(bns, before):
            now = m.__dict__.get('_rdfc_pending_batches', 0)
            if now != b:
                self._bn_counts.append((m, now - b))
                m.__dict__['_rdfc_pending_batches'] = b
        Fh.invalidate_packed_weights()
        self._graphs = (g1, g2, g3, stats)

    def _optimize_graphed(self, sync_losses):
        self._draw_pepper_static()
        for o in self.optimizers.values():
            o.sync_hyper()
        if self._graphs is None:
            try:
                self._capture()
            except Exception as e:              # capture is an optimisation: a failed one must not take the run down
                import warnings
                warnings.warn(f'HIP graph capture failed ({type(e).__name__}: {e}); continuing with eager launches')
                torch.cuda.synchronize()
                self._graph_mode, self._graphs = False, None
                self.set_input(self._batch)
                return self.optimize_parameters(sync_losses)
        g1, g2, g3, stats = self._graphs
        for o in self.optimizers.values():
            o.begin_step()                  # (a replay runs no Python: the buckets' per-step state is reset here)
        g1.replay()
        self.optimizer_G.all_reduce_grads_async()
        g2.replay()
        self.optimizer_D.all_reduce_grads_async()
        self.optimizer_G.wait_grads()
        self.optimizer_D.wait_grads()
        g3.replay()
        for o in self.optimizers.values():
            o.steps += 1
        for m, k in self._bn_counts:
            m.__dict__['_rdfc_pending_batches'] += k
        return self.reduce_loss(stats) if sync_losses else stats

    def optimize_parameters(self, sync_losses=True):
        if getattr(self, '_graph_mode', False):
            return self._optimize_graphed(sync_losses)
        loss_stats = {}
        rb = None
        if sync_losses:
            rb = self.__dict__.get('_rb')
            if rb is None or rb.device != self.device:
                rb = self._rb = _LossReadback(self.device)
            # (the default schedule posts G's, D_B's and D_A's losses before any gradient bucket of the step goes out: see _LossReadback)
            early = self._batched() and _D_UNDER_G == 'first' and Fh.multi_stream() and self.device.type == 'cuda'
            rb.begin(expect=3 if early else None)
        self._rb_active = rb
        try:
            self._enqueue_step(loss_stats)
        finally:
            self._rb_active = None
        return rb.collect(list(loss_stats)) if sync_losses else loss_stats

    def _enqueue_D_steps(self, loss_stats):
        """Both discriminator steps on a helper stream, forked BEFORE G's backward: they depend on the forward only (fake_A /
        fake_B detached; the G-loss passes through D_A / D_B were built with the discriminators frozen, so G's backward writes
        none of their gradients), and their small kernels (27x37 maps) fill a quarter of the chip when they run alone behind
        G's backward -- here they run beside it.  FlatAdam.step() joins the streams."""
        self.set_requires_grad([self.disc_A, self.disc_B], True)
        main = torch.cuda.current_stream(self.device)
        first = _D_UNDER_G == 'first'
        # enqueued before G's backward they take the depth-branch stream (idle until G_A2B's backward reaches that branch);
        # behind it the B->A stream (G's backward has used it for D_B's input gradient by then)
        stream = Fh.branch_stream(self.device) if first else Fh.net_stream(self.device)
        stream.wait_event(self._fork_D)
        with torch.cuda.stream(stream):
            self.optimizer_D.zero_grad()
            stats_DB = self.backward_D_B2A()
            stats_DA = self.backward_D_A2B()
            self.optimizer_D.all_reduce_bucket_async(1, helpers=False)
            self.optimizer_D.all_reduce_bucket_async(0, helpers=False)
        for t in list(stats_DB.values()) + list(stats_DA.values()):
            t.record_stream(main)
        self._D_stats = dict(stats_DA)
        self._D_stats.update(stats_DB)
        self._D_done = True
        if first:
            self.set_requires_grad([self.disc_A, self.disc_B], False)        # (host-side flag only; G's graph is built already)

    def _enqueue_step(self, loss_stats):
        self._D_done, self._D_stats = False, None
        self._D_hook = (lambda: self._enqueue_D_steps(loss_stats)) if self._batched() else None
        self.forward()
        self.set_requires_grad([self.disc_A, self.disc_B], False)
        self.optimizer_G.zero_grad()
        if self.G_normal is not None:
            self.optimizer_Normal.zero_grad()
        if self.use_label_net:
            self.optimizer_Label.zero_grad()
        loss_stats.update(self.backward_G())
        # G gradients fly over xGMI on the side stream while the D step computes
        self.optimizer_G.all_reduce_grads_async()
        if self.G_normal is not None:
            self.optimizer_Normal.all_reduce_grads_async()
        if self.use_label_net:
            self.optimizer_Label.all_reduce_grads_async()
        self.set_requires_grad([self.disc_A, self.disc_B], True)
        fork = self.__dict__.get('_fork_D')
        if fork is not None:
            if not self.__dict__.get('_D_done'):
                self._enqueue_D_steps(loss_stats)
            loss_stats.update(self._D_stats)
        else:
            self.optimizer_D.zero_grad()
            with self._net_stream_ctx():              # the two discriminators are independent: D_B's step on the second stream
                stats_DB = self.backward_D_B2A()
                self.optimizer_D.all_reduce_bucket_async(1)       # D_B's bucket: under D_A's backward
            loss_stats.update(self.backward_D_A2B())
            self._join_net_stream(tuple(stats_DB.values()))
            loss_stats.update(stats_DB)
            self.optimizer_D.all_reduce_grads_async()             # D_A's bucket: under G's Adam update
        self.optimizer_G.step()
        if self.G_normal is not None:
            self.optimizer_Normal.step()
        if self.use_label_net:
            self.optimizer_Label.step()
        self.optimizer_D.step()
        if self.scaler is not None:
            self.scaler.update()
        # nothing of this step's autograd graph may outlive it (a later HIP-graph capture builds its own nodes: _capture)
        self._D_hook = self._D_stats = self._fork_D = None

    def reduce_loss(self, loss):
        """base.py:121-132 semantics (mean over ranks, python floats) with one collective and one D2H copy"""
        keys = list(loss)
        packed = torch.stack([loss[k].detach().float().reshape(()) for k in keys])
        if dp_active():
            packed = packed / dist.get_world_size()
            if not optim.COLLECTIVES_OFF:
                dist.all_reduce(packed)
        return dict(zip(keys, packed.tolist()))

    # ---- mode / lr / checkpoints (base.py:74-117; lib/utils/checkpoint.py:21-58,113-166) -----------------------
    def eval(self):
        for m in self.models.values():
            m.eval()

    def train(self):
        for m in self.models.values():
            m.train()

    def update_learning_rate(self):
        for s in self.lr_schedulers.values():
            s.step()

    def save_ckpt(self, filename, meta=None):
        ckpt = dict(meta=meta or {})
        for name, m in self.models.items():
            ckpt[f'{name}_state_dict'] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        if self.is_train:
            for name, o in self.optimizers.items():
                ckpt[f'{name}_optimizer_state_dict'] = o.state_dict()
            for name, s in self.lr_schedulers.items():
                ckpt[f'{name}_lr_scheduler_state_dict'] = s.state_dict()
            if self.scaler is not None:           # (a key of its own: upstream-format checkpoints simply do not have it)
                ckpt['loss_scaler_state_dict'] = self.scaler.state_dict()
        torch.save(ckpt, filename)

    def load_from(self, load_path, strict=False):
        ckpt = torch.load(load_path, map_location='cpu')
        for name, m in self.models.items():
            if f'{name}_state_dict' in ckpt:
                sd = {k[7:] if k.startswith('module.') else k: v for k, v in ckpt[f'{name}_state_dict'].items()}
                m.load_state_dict(sd, strict=strict)
        Fh.invalidate_packed_weights()
        return ckpt

    def resume_from(self, resume_path):
        ckpt = self.load_from(resume_path, strict=True)
        for name, o in self.optimizers.items():
            if f'{name}_optimizer_state_dict' in ckpt:
                o.load_state_dict(ckpt[f'{name}_optimizer_state_dict'])
        for name, s in self.lr_schedulers.items():
            if f'{name}_lr_scheduler_state_dict' in ckpt:
                s.load_state_dict(ckpt[f'{name}_lr_scheduler_state_dict'])
        if self.scaler is not None and 'loss_scaler_state_dict' in ckpt:
            self.scaler.load_state_dict(ckpt['loss_scaler_state_dict'])
        return ckpt['meta'].get('epoch', 0)

    def save_samples(self, save_dir, iters):
        """visualisation dumps are outside the hot path (rdfc_gan.py:342-410)"""
        return None
