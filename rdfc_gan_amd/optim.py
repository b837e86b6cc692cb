"""Flat-buffer Adam + data-parallel gradient exchange for the hot-path nets.

torch.optim.Adam semantics (rdfc_gan.py:135-140: lr, betas=(beta1, beta2), eps 1e-8, no
weight decay) over ONE contiguous f32 buffer per optimizer, so that
  * the update is one HIP kernel launch (rdfc_adam_step) instead of ~200 small ones,
  * the data-parallel gradient exchange is one RCCL all-reduce per optimizer on a side
    HIP stream (G: 168 MB, D: 22 MB), overlapped with whatever the step does next.
Parameters keep their nn.Parameter identity and state_dict keys: their storage is
re-pointed into the flat buffer, and `.grad` is a persistent view of the flat gradient
buffer (autograd accumulates into it in place).
"""
import os

import numpy as np

import torch
import torch.distributed as dist

from . import functional as Fh
from . import hip


def _pview(buf, off, p):
    """view of p's slice of a flat buffer with p's logical shape; conv weights (4-D) are stored channels-last
    ([O][KH][KW][I] in memory), the layout the weight-gradient kernels write and the weight packer reads."""
    k = p.numel()
    if p.dim() == 4:
        O, I, KH, KW = p.shape
        return buf[off:off + k].view(O, KH, KW, I).permute(0, 3, 1, 2)
    return buf[off:off + k].view(p.shape)


class LossScaler:
    """Loss scale of the f16 mode, kept on the device and shared by the optimizers of one model (include/rdfc_hip.h, "Dynamic
    loss scaling").  Every backward starts from the cotangent `seed` (a 0-dim view of the device value); FlatAdam.step() checks
    its all-reduced gradients for inf / NaN, skips the update when it finds one, and `update()` -- once per train step, behind
    the optimizers -- halves the scale after an overflow and doubles it after `interval` clean steps (torch.cuda.amp.GradScaler's
    rule).  dynamic=False: the scale stays put, an overflowing step is still skipped instead of poisoning Adam's moments and
    the master weights.  Nothing here makes the host wait; `scale()` / `skipped()` read the device values back (reporting only)."""

    def __init__(self, device, init_scale=1024.0, dynamic=True, growth=2.0, backoff=0.5, interval=2000, min_scale=1.0,
                 max_scale=2.0 ** 24):
        self.dynamic, self.growth, self.backoff, self.interval = bool(dynamic), float(growth), float(backoff), int(interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        self.state = torch.tensor([float(init_scale), 1.0 / float(init_scale), 0.0, 0.0], dtype=torch.float32, device=device)
        self._optimizers = []

    @property
    def seed(self):
        return self.state[0]

    def attach(self, optimizer):
        optimizer.scaler = self
        optimizer._ostate = torch.zeros(2, dtype=torch.float32, device=self.state.device)
        self._optimizers.append(optimizer)

    def update(self):
        g, b = (self.growth, self.backoff) if self.dynamic else (1.0, 1.0)
        hip.loss_scale_update(self.state, g, b, self.interval, self.min_scale, self.max_scale)

    def scale(self):
        return float(self.state[0])

    def set_scale(self, value):
        self.state[:2] = torch.tensor([float(value), 1.0 / float(value)], dtype=torch.float32)

    def skipped(self):
        """steps each attached optimizer has skipped so far (host read-back)"""
        return [int(o._ostate[1]) for o in self._optimizers]

    # ---- checkpointing: a resumed f16 run continues with the scale it had reached, not with the initial one -----------------
    def state_dict(self):
        st = [float(v) for v in self.state.cpu()]
        return dict(scale=st[0], clean_steps=int(st[2]), skipped=self.skipped())

    def load_state_dict(self, sd):
        """call AFTER the optimizers' load_state_dict (which stores the steps really taken and zeroes the skip counters): restores
        the scale, the clean-step counter and every optimizer's skip counter (its step count grows by the same number, so the
        bias corrections -- over steps minus skipped -- continue unchanged)"""
        scale = float(sd['scale'])
        self.state.copy_(torch.tensor([scale, 1.0 / scale, float(sd.get('clean_steps', 0)), 0.0], dtype=torch.float32))
        for o, k in zip(self._optimizers, sd.get('skipped', [])):
            k = int(k)
            o._ostate.copy_(torch.tensor([0.0, float(k)], dtype=torch.float32))
            o.steps += k
            if o._hyper is not None:
                o._hyper[1] = float(o.steps)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.5, 0.999), eps=1e-8, weight_decay=None):
        """weight_decay=None: torch.optim.Adam (rdfc_gan.py:135-140); a number: torch.optim.AdamW with that decoupled decay
        (upstream's optimizer_Normal, :141-144).  Which of the two update rules runs is fixed here, at construction
        (`self._adamw`); `param_groups[0]['weight_decay']` always holds a number (0 for Adam), as torch.optim.Adam's own
        state_dict does, so that checkpoints go both ways and a loaded state_dict cannot switch the kernel."""
        params = [p for p in params]
        assert params, 'optimizer got an empty parameter list'
        self._adamw = weight_decay is not None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=float(weight_decay or 0.0)))
        dev = params[0].device
        self._params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                v = _pview(self.flat, off, p)
                v.copy_(p.detach())
                p.data = v
                p.grad = _pview(self.flat_grad, off, p)
                off += k
        self.steps = 0
        self.loss_scale = 1.0         # the gradients in flat_grad are this many times too large (a fixed host-side factor)
        self.scaler = None            # LossScaler.attach(): device-side (dynamic) loss scale + overflow check, replaces loss_scale
        self._ostate = None           # device {overflow in this step, steps skipped}
        self._pending = []            # events of the in-flight bucket all-reduces
        self._buckets = [[(0, n)]]
        self._started = set()
        self.started_order = []       # bucket indices in the order their exchange was started in the current step (tests read it)
        self._hyper = None            # device {lr, step}: set by make_capturable(); the step then launches rdfc_adam_step_dev
        self._hyper_lr = None
        self.host_counts = True       # False while a captured graph replays step(): the caller advances self.steps
        Fh.invalidate_packed_weights()

    def make_capturable(self):
        """learning rate and step count move to device memory so that step() can be captured in a HIP graph"""
        if self._hyper is None:
            self._hyper_lr = float(self.param_groups[0]['lr'])
            self._hyper = torch.tensor([self._hyper_lr, float(self.steps)], dtype=torch.float32, device=self.flat.device)

    def sync_hyper(self):
        """host -> device refresh of what a scheduler or load_state_dict may have changed (eager; call before a replay)"""
        if self._hyper is not None:
            lr = float(self.param_groups[0]['lr'])
            if lr != self._hyper_lr:
                self._hyper[0] = lr
                self._hyper_lr = lr

    def begin_step(self):
        """per-step bookkeeping of the bucket exchange that must also run when zero_grad() does not -- a replayed HIP graph runs
        the captured zero-fill but no Python: without this every bucket still counts as started from the capture on and the
        replicas would stop exchanging gradients"""
        self._started = set()
        self.started_order = []

    def zero_grad(self, set_to_none=False):
        # (no join of the helper streams here: step() / the all-reduce joined them before they read the gradients, and the
        # other optimizer's weight gradients may keep running under this one's backward)
        self.wait_grads()
        self._started = set()
        self.started_order = []
        if self.flat_grad.is_cuda:
            hip.zero_(self.flat_grad)       # hipMemsetAsync
        else:
            self.flat_grad.zero_()
        for p in self._params:        # autograd may have replaced a view when grads were None
            if p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_grad.data_ptr() + self.flat_grad.numel() * 4:
                self._rebind()
                break

    def _rebind(self):
        off = 0
        for p in self._params:
            p.grad = _pview(self.flat_grad, off, p)
            off += p.numel()

    # ---- data parallel: buckets of the flat gradient, side stream ---------------------------------------------------
    # The flat gradient is cut into buckets (element ranges) in the order in which they become final during backward;
    # the train step starts a bucket's sum-all-reduce (RCCL over xGMI, on a communication stream of its own) as soon as
    # the backward of the network that owns it has been enqueued, so that the exchange runs under the rest of the
    # backward / the next phase of the step.  1/world_size is folded into the Adam kernel.  Default: one bucket.
    def set_buckets(self, buckets):
        """buckets: [bucket, ...]; a bucket is an element range (start, end) of the flat buffers or a list of such ranges (the
        parameters whose gradients become final together need not be neighbours in the flat buffer: RDFGenerator's decoders sit between
        its encoders in parameter order).  All ranges disjoint, covering everything."""
        norm = []
        for b in buckets:
            rs = [b] if isinstance(b[0], (int, np.integer)) else list(b)
            rs = sorted((int(a), int(c)) for a, c in rs if c > a)
            merged = []
            for a, c in rs:                       # neighbours become one collective
                if merged and merged[-1][1] == a:
                    merged[-1] = (merged[-1][0], c)
                else:
                    merged.append((a, c))
            if merged:
                norm.append(merged)
        flat = sorted(r for b in norm for r in b)
        assert flat[0][0] == 0 and flat[-1][1] == self.flat.numel()
        assert all(x[1] == y[0] for x, y in zip(flat, flat[1:])), 'buckets must tile the flat buffer'
        self._buckets = norm
        self._started = set()

    def bucket_ranges(self, i):
        return list(self._buckets[i])

    def param_range(self, params):
        """element range of a run of consecutive parameters (as passed to the constructor) in the flat buffers"""
        ids = {id(p) for p in params}
        off, lo, hi = 0, None, None
        for p in self._params:
            if id(p) in ids:
                lo = off if lo is None else lo
                hi = off + p.numel()
            off += p.numel()
        return lo, hi

    @staticmethod
    def _distributed():
        return dp_active()

    def all_reduce_bucket_async(self, i, helpers=True, after=()):
        """start the all-reduce of bucket i (idempotent within a step).  Nothing on the compute streams waits here: the
        collective is issued (async_op) from the weight-gradient stream after that stream has been made to wait for the
        current one and the other helper streams -- its later launches wait for the current stream anyway -- so the process
        group's own stream picks the dependencies up and no further HIP stream is needed.  (The runtime multiplexes streams
        onto GPU_MAX_HW_QUEUES = 4 hardware queues and the step already uses four: every further stream shares a queue with
        a compute stream and its waits stall that stream.  One-rank RCCL runs of bench.py: 38.7 ms per step without a process
        group, 41.3 with one and a communication stream per optimizer, 40.3-40.9 once the compute streams are bound to their
        queues first -- functional.reserve_streams -- with either scheme)
        helpers=False: the bucket's gradients come from the current stream and the weight-gradient stream only (the
        discriminators' step beside G's backward): do not wait for what the other helper streams are still running.
        after: events the collective must also follow -- a bucket started from inside a backward is started by the LAST of several
        markers, each of which fired on its own stream; the marker that fired first recorded an event there (round-4 advice: the
        issue stream otherwise follows only the last marker's stream, and gradients that autograd accumulates on the first
        marker's stream -- W-AdaIN's style weights on the main stream -- could still be in flight when the collective reads them)."""
        if not self._distributed() or i in self._started:
            return
        self._started.add(i)
        self.started_order.append(i)
        views = [self.flat_grad[a:b] for a, b in self._buckets[i]]
        if COLLECTIVES_OFF:                   # diagnostic (bench.py's comm block): the step's control flow without the exchange
            return
        if not self.flat_grad.is_cuda:
            for view in views:
                dist.all_reduce(view)         # gloo / CPU tests
            return
        dev = self.flat_grad.device
        cur = torch.cuda.current_stream(dev)
        issue = Fh.wgrad_stream(dev) or cur
        if issue != cur:
            issue.wait_stream(cur)
        for e_ in after:
            issue.wait_event(e_)
        for s_ in (Fh.helper_streams(dev) if helpers else []):
            if s_ != issue and s_ != cur:
                issue.wait_stream(s_)
        with torch.cuda.stream(issue):
            ready = None
            if COMM_TRACE is not None:        # when everything the bucket depends on has run: the earliest the exchange can start
                ready = torch.cuda.Event(enable_timing=True)
                ready.record()
            for view in views:
                self._pending.append((i, ready, dist.all_reduce(view, async_op=True)))

    def all_reduce_grads_async(self):
        """start the all-reduce of every bucket that has not been started in this step.  No-op when not distributed."""
        if not self._distributed():
            return
        for i in range(len(self._buckets)):
            self.all_reduce_bucket_async(i)

    def wait_grads(self):
        """the current stream waits for the bucket all-reduces started so far.  Under COMM_TRACE every bucket's wait sits between two
        events on the waiting stream: what that stream stalls for is the exposed (not overlapped) part of the bucket's exchange."""
        if COMM_TRACE is None:
            for _i, _ready, w in self._pending:
                w.wait()
        else:
            k = 0
            while k < len(self._pending):
                i, ready = self._pending[k][:2]
                before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                before.record()
                while k < len(self._pending) and self._pending[k][0] == i:
                    self._pending[k][2].wait()
                    k += 1
                after.record()
                COMM_TRACE.append(dict(optimizer=self, bucket=i, ready=ready, wait_begin=before, wait_end=after))
        self._pending = []

    @torch.no_grad()
    def step(self, closure=None):
        Fh.join_side_stream()
        self.wait_grads()
        g = self.param_groups[0]
        nranks = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        world = nranks * float(self.loss_scale)
        if not self.flat.is_cuda:
            raise RuntimeError('FlatAdam.step runs the HIP kernel rdfc_adam_step: parameters must live on the GPU '
                               '(there is no CPU fallback)')
        if self._adamw and self._hyper is not None:
            raise RuntimeError('the AdamW update is not capturable (host-side lr / step arguments): graph mode covers Adam only')
        if self.scaler is not None:
            # f16 mode: the (all-reduced, so identical on every rank) gradients are checked on the device; an overflow skips the
            # update as a whole and LossScaler.update() backs the scale off
            hip.grad_nonfinite(self.flat_grad, self._ostate)
            if self._hyper is not None:
                self._hyper[1:2] += 1.0
            if self._hyper is None or self.host_counts:
                self.steps += 1
            hip.adam_step_scaled(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                                 self.steps, self.scaler.state, self._ostate, grad_scale=1.0 / world,
                                 weight_decay=float(g.get('weight_decay') or 0.0) if self._adamw else 0.0, hyper=self._hyper)
        elif self._adamw:
            self.steps += 1
            hip.adamw_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                           float(g.get('weight_decay') or 0.0), self.steps, 1.0 / world)
        elif self._hyper is not None:
            self._hyper[1:2] += 1.0           # device-side step count (captured with the update)
            hip.adam_step_dev(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self._hyper, g['betas'][0],
                              g['betas'][1], g['eps'], 1.0 / world)
            if self.host_counts:
                self.steps += 1
        else:
            self.steps += 1
            hip.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1],
                          g['eps'], self.steps, 1.0 / world)
        Fh.weights_updated(self.flat)

    def steps_taken(self):
        """optimizer steps that really updated the parameters (steps minus those skipped for an overflow; reads the device)"""
        return self.steps - (int(self._ostate[1]) if self._ostate is not None else 0)

    # ---- checkpoint format of torch.optim.Adam (lib/utils/checkpoint.py:21-58 stores optimizer.state_dict()) ----
    def state_dict(self):
        state, off = {}, 0
        taken = self.steps_taken()
        for i, p in enumerate(self._params):
            k = p.numel()
            state[i] = dict(step=torch.tensor(float(taken)),
                            exp_avg=_pview(self.exp_avg, off, p).clone(memory_format=torch.contiguous_format),
                            exp_avg_sq=_pview(self.exp_avg_sq, off, p).clone(memory_format=torch.contiguous_format))
            off += k
        g = dict(self.param_groups[0])
        g['params'] = list(range(len(self._params)))
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        off = 0
        for i, p in enumerate(self._params):
            k = p.numel()
            st = sd['state'].get(i)
            if st is not None:
                _pview(self.exp_avg, off, p).copy_(st['exp_avg'])
                _pview(self.exp_avg_sq, off, p).copy_(st['exp_avg_sq'])
                self.steps = int(st['step'])
            off += k
        if self._ostate is not None:
            self._ostate.zero_()
        if self._hyper is not None:
            self._hyper[1] = float(self.steps)
        for k_, v in sd['param_groups'][0].items():
            if k_ == 'weight_decay':            # a number in every torch checkpoint; None only in round-1 files of this class
                v = float(v or 0.0)
                if not self._adamw and v != 0.0:
                    raise ValueError('checkpoint carries weight_decay != 0 for an optimizer built as plain Adam '
                                     '(coupled L2 decay is not implemented by rdfc_adam_step)')
            if k_ != 'params':
                self.param_groups[0][k_] = v




# diagnostics of the data-parallel exchange (bench.py's `comm` block, tests/test_dp*.py): COMM_TRACE = [] collects, per started bucket,
# HIP events for "dependencies met" (issue stream) and the waiting stream's stall for it; COLLECTIVES_OFF = True runs the very same step
# without the collectives (gradient buckets and loss means stay local) -- what the exchange costs the step is the difference.
COMM_TRACE = None
COLLECTIVES_OFF = False


def dp_active():
    """whether the step exchanges gradients / losses: a process group with more than one rank -- or any initialised group
    under RDFC_DP_FORCE=1, which sends a one-rank job through the very same collectives (how the RCCL path is exercised on a
    one-GPU box: tests/test_dp_gpu.py, bench.py --rccl-self-test)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('RDFC_DP_FORCE') == '1'


def broadcast_module_state(modules, src=0):
    """DDP-constructor semantics: every rank starts from rank `src`'s parameters and buffers."""
    if not dp_active():         # (RDFC_DP_FORCE=1: a one-rank job broadcasts through the backend all the same)
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)
    Fh.invalidate_packed_weights()


def resync_replicas(modules, optimizers, src=0):
    """Make every rank a copy of rank `src` again: parameters (the optimizers' flat buffers: the parameters are views of them), module
    buffers (packed into one tensor per dtype: a few hundred BatchNorm statistics one broadcast each cost a gloo job a minute), Adam
    moments, step counts and the device-side loss-scaler state of every optimizer.  For diagnostics that let replicas drift on purpose
    (bench.py's collectives-off leg runs optimizer steps on local gradients); training never needs it."""
    if not dp_active():
        return
    seen = set()
    for o in optimizers:
        for t in (o.flat, o.exp_avg, o.exp_avg_sq, o._ostate, o._hyper, getattr(o.scaler, 'state', None)):
            if t is not None and id(t) not in seen:
                seen.add(id(t))
                dist.broadcast(t, src)
        steps = torch.tensor([o.steps], dtype=torch.int64, device=o.flat.device)
        dist.broadcast(steps, src)
        o.steps = int(steps)
    covered = {p.data_ptr() for o in optimizers for p in o._params}
    rest = [t for m in modules for t in list(m.parameters()) + list(m.buffers()) if t.data_ptr() not in covered and t.numel() > 0]
    by_dtype = {}
    for t in rest:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt_, ts in by_dtype.items():
        packed = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(packed, src)
        off = 0
        for t in ts:
            t.data.copy_(packed[off:off + t.numel()].view_as(t))
            off += t.numel()
    Fh.invalidate_packed_weights()
