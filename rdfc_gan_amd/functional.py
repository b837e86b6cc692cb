"""Autograd blocks over the HIP kernels.

Every block is one torch.autograd.Function whose forward/backward are sequences of
C-ABI calls (rdfc_gan_amd.hip).  Activations flowing between blocks are pixel-major
maps [N,H,W,C] in the compute dtype (bf16 or f32); module inputs / outputs are
converted from / to upstream's NCHW f32 at the network boundary only.

Blocks
  conv_block   conv | conv-transpose  [+bias] [+BatchNorm | InstanceNorm] [+residual] [+act]
  wadain       W-AdaIN fusion (per-pixel style affine of an instance-normalised map)
  crop_cat     upstream's `_concat` (crop trailing rows/cols, concatenate channels)
  reflect_pad, to_pm / to_nchw, fusion_head
"""
import math
import os
import weakref

import torch

from . import hip

# Normalisation reductions are two-stage (per-workgroup partials + a fixed-order f64 finalize launch): bitwise reproducible.
# (A one-stage variant -- f64 atomics into one accumulator per (group, channel), no finalize launches -- was built in round 1
# and measured SLOWER on MI355X: ~540 workgroups adding to the same address serialise at ~54 ns per add, statistics kernels
# 1.6 -> 6.4 ms and backward reductions 4.4 -> 9.2 ms per step against 2.2 ms of finalize launches saved; removed in round 3.)
_compute_dtype = torch.bfloat16
_weights_epoch = 0


def set_compute_dtype(dtype):
    """torch.bfloat16 (fast path: bf16 MFMA), torch.float16 (f16 MFMA; BASELINE config "fp16 activations with fp32 loss
    accumulate" -- pair it with RDFCGAN(loss_scale=...), half precision underflows on unscaled gradients) or torch.float32
    (exact-f32 MFMA, parity mode)."""
    global _compute_dtype
    assert dtype in (torch.bfloat16, torch.float16, torch.float32)
    _compute_dtype = dtype


def compute_dtype():
    return _compute_dtype


def invalidate_packed_weights():
    """call after parameters were changed behind autograd's back (load_state_dict, an optimizer that is not FlatAdam): every
    packed operand is stale and is rebuilt at its next use."""
    global _weights_epoch
    _weights_epoch += 1
    for k in [k for k, e in _pack_reg.items() if e.ref() is None]:
        del _pack_reg[k]
    _pack_tables.clear()


class _Pack:
    """the packed operands (w_fwd, w_bwd) of one weight tensor, kept across steps"""
    __slots__ = ('ref', 'ptr', 'wf', 'wb', 'epoch', 'version', 'scale', 'shape', 'dtype', 'strides')


_pack_reg = {}          # (data_ptr, dtype, scale) -> _Pack
_pack_tables = {}       # (lo, hi, dtype) -> (registry generation, device job table, njobs, nblocks, entries)
_pack_generation = 0    # bumped whenever an entry is added or replaced
pack_misses = 0         # single-weight pack launches so far (diagnostic: a steady-state step has none)


def _packed(weight, dtype, scale):
    global _pack_generation
    key = (weight.data_ptr(), dtype, scale)
    e = _pack_reg.get(key)
    # identity of the memory behind the key: the tensor the weight is a view of (W-AdaIN views its linear weight as a 1x1 conv
    # weight afresh in every call) or the weight itself; a dead or different owner means the address has been recycled
    owner = weight._base if weight._base is not None else weight
    if e is not None and e.ref() is owner and e.epoch == _weights_epoch and e.version == weight._version \
            and e.shape == weight.shape:
        return e.wf, e.wb
    global pack_misses
    pack_misses += 1
    w = weight.detach()
    e = _Pack()
    if hip.is_ohwi(w) and not w.is_contiguous():    # channels-last master weights (FlatAdam): no OIHW copy
        e.wf, e.wb = hip.pack_weight_ohwi(w, dtype, scale)
    else:
        w = w.contiguous()
        e.wf, e.wb = hip.pack_weight(w, dtype, scale)
    e.ref, e.ptr, e.epoch, e.version = weakref.ref(owner), weight.data_ptr(), _weights_epoch, weight._version
    e.scale, e.shape, e.dtype = scale, weight.shape, dtype
    e.strides = tuple(weight.stride()) if w.data_ptr() == weight.data_ptr() else None     # None: packed from a copy
    _pack_reg[key] = e
    _pack_generation += 1
    return e.wf, e.wb


def weights_updated(flat):
    """FlatAdam.step() changed the weights stored in `flat` (behind autograd's back, on the current stream, every helper stream
    joined): rebuild the packed operands the step has been using, all in one launch (rdfc_pack_weights_batch) into the buffers
    they already have -- instead of ~80 small launches at their next first uses.  Operands outside `flat` stay as they are."""
    if not flat.is_cuda or torch.cuda.is_current_stream_capturing():
        invalidate_packed_weights()
        return
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
    for dtype in {e.dtype for e in _pack_reg.values() if lo <= e.ptr < hi}:
        tab = _pack_tables.get((lo, hi, dtype))
        if tab is None or tab[0] != _pack_generation:
            ents = [e for e in _pack_reg.values() if lo <= e.ptr < hi and e.dtype == dtype and e.strides is not None
                    and e.epoch == _weights_epoch and e.ref() is not None and e.version == e.ref()._version]
            tab = (_pack_generation,) + hip.pack_job_table(ents, flat.device) + (ents,)
            _pack_tables[(lo, hi, dtype)] = tab
        _, jobs, njobs, nblocks, ents = tab
        if njobs:
            hip.pack_weights_batch(jobs, njobs, nblocks, dtype)
    # whatever of this buffer is not in a table (packed from a copy, stale already) is rebuilt at its next use
    for e in _pack_reg.values():
        if lo <= e.ptr < hi and e.strides is None:
            e.epoch = -1


def _accumulable(g):
    return g is not None and g.dtype == torch.float32 and g.is_contiguous() and not g.requires_grad


# ---- weight gradients on a side stream ----------------------------------------------------------------------------
# A weight gradient that accumulates straight into .grad has no consumer before the optimizer step, so it does not
# have to sit in the backward chain: it is launched on a second HIP stream (forked behind the kernel that produced dy)
# and the MFMA-bound split-K GEMM overlaps the HBM-bound normalisation kernels of the layers below.  join_side_stream()
# brings the streams back together; FlatAdam calls it before it touches the gradients.  RDFC_WGRAD_STREAM=0 disables it.
_WGRAD_STREAM = os.environ.get('RDFC_WGRAD_STREAM', '1') != '0'
_multi_stream = True        # False while a step is captured into / replayed from HIP graphs (single-stream capture)


def set_multi_stream(enabled):
    """helper streams (weight gradients, RDFGenerator's depth branch, the B->A half of the G step) on / off"""
    global _multi_stream
    join_side_stream()
    _multi_stream = bool(enabled)


def multi_stream():
    return _multi_stream


# The same parameter is used on two HIP streams within a step by design (G_A2B on the main and on the B->A stream): autograd
# orders the two gradient accumulations with an event wait, which is what is wanted; its warning about the mismatch is noise here.
if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
_side = {}          # device -> [stream, list of tensors the pending launches still read]

def _side_stream(device):
    s = _side.get(device)
    if s is None:
        s = _side[device] = [torch.cuda.Stream(device=device), []]
    return s


_branch = {}


def branch_stream(device):
    """second compute stream for an independent branch of a network (RDFGenerator's depth branch)"""
    s = _branch.get(device)
    if s is None:
        s = _branch[device] = torch.cuda.Stream(device=device)
    return s


_net = {}


def net_stream(device):
    """third compute stream: the half of the train step that does not depend on the other (RDFCGAN: the B->A chain)"""
    s = _net.get(device)
    if s is None:
        s = _net[device] = torch.cuda.Stream(device=device)
    return s


def wgrad_stream(device):
    """the weight-gradient stream of `device` when weight gradients are launched on one (None otherwise)"""
    if not (_WGRAD_STREAM and _multi_stream) or torch.device(device).type != 'cuda':
        return None
    return _side_stream(torch.device(device))[0]


def reserve_streams(device, with_net=False):
    """create the step's HIP streams NOW and put one launch on each.  The HIP runtime binds a stream to one of its
    GPU_MAX_HW_QUEUES (4) hardware queues at the stream's first use, round-robin; the batched step uses three (default, weight
    gradients, depth branch / discriminator step).  Called before torch.distributed.init_process_group -- whose RCCL communicator
    brings a stream of its own -- it gives the three compute streams a queue each and leaves the fourth to RCCL; called after,
    two compute streams can end up sharing one (a wait queued on one then stalls the other).  with_net: also the B->A stream,
    which the step uses for D_B's G-loss pass when no process group exists (worth 0.4 ms per step)."""
    device = torch.device(device)
    if device.type != 'cuda':
        return
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    for s in (torch.cuda.default_stream(device), _side_stream(device)[0], branch_stream(device)) + ((net_stream(device),) if with_net else ()):
        with torch.cuda.stream(s):
            torch.zeros(64, device=device)
    torch.cuda.synchronize(device)


def helper_streams(device):
    """the helper streams that exist on `device` (weight-gradient, branch, net): what a consumer of the gradients on yet
    another stream has to wait for"""
    out = []
    side = _side.get(device)
    if side is not None:
        out.append(side[0])
    for tab in (_branch, _net):
        if tab.get(device) is not None:
            out.append(tab[device])
    return out


def join_side_stream():
    """make the current stream wait for everything launched on the helper streams (weight gradients on the side stream,
    the branch / net streams they may have been forked from); release the tensors kept alive for them"""
    for dev in set(_side) | set(_branch) | set(_net):       # (RDFC_WGRAD_STREAM=0 leaves _side empty: the others still join)
        cur = torch.cuda.current_stream(dev)
        for extra in (_branch.get(dev), _net.get(dev)):
            if extra is not None and extra != cur:
                cur.wait_stream(extra)
        side = _side.get(dev)
        if side is not None and side[1]:
            cur.wait_stream(side[0])
            side[1].clear()


def _wgrad(d, a, b, weight, O, I, k, scale, reflect=False):
    """weight gradient of one conv; -> tensor for autograd, or None after accumulating into weight.grad in place.

    rdfc_conv2d_wgrad produces [O][k*k][I], which IS the memory of a channels-last OIHW tensor.  When the parameter
    and its persistent .grad are stored that way (FlatAdam does it) the kernel accumulates straight into .grad:
    no zero-fill, no unpack, no autograd add."""
    g = weight.grad
    if scale == 1.0 and g is not None and g.dtype == torch.float32 and hip.is_ohwi(g) and not g.requires_grad:
        if _WGRAD_STREAM and _multi_stream and a.is_cuda and hip.PROFILE is None:
            stream, keep = _side_stream(a.device)
            stream.wait_stream(torch.cuda.current_stream(a.device))
            hip.conv_wgrad(d, a, b, into=g, stream=stream.cuda_stream, reflect=reflect)      # launched straight onto the side stream
            keep.append((a, b))         # their memory must not be recycled before the join
        elif _WGRAD_STREAM and _multi_stream and a.is_cuda:
            stream, keep = _side_stream(a.device)
            stream.wait_stream(torch.cuda.current_stream(a.device))
            with torch.cuda.stream(stream):                                 # (per-kernel events record on the current stream)
                hip.conv_wgrad(d, a, b, into=g, reflect=reflect)
            keep.append((a, b))
        else:
            hip.conv_wgrad(d, a, b, into=g, reflect=reflect)
        return None
    if scale == 1.0:
        dw = hip.conv_wgrad(d, a, b, reflect=reflect)
        return dw.view(O, k, k, I).permute(0, 3, 1, 2)
    return hip.unpack_wgrad(hip.conv_wgrad(d, a, b, reflect=reflect), O, I, k, k, scale)


class ConvSpec:
    """static description of one conv block (shared by forward and backward)"""
    __slots__ = ('k', 'stride', 'pad', 'transposed', 'out_pad', 'norm', 'act', 'slope', 'eps', 'momentum',
                 'out_f32', 'wscale', 'dil', 'group_momenta', 'passthrough', 'reflect')

    def __init__(self, k, stride=1, pad=0, transposed=False, out_pad=0, norm='none', act='none', slope=0.0,
                 eps=1e-5, momentum=0.1, out_f32=False, wscale=1.0, dil=1, group_momenta=None, passthrough=False, reflect=0):
        """group_momenta (BatchNorm only): the batch is G = len(group_momenta) stacked batches of as many separate calls of
        this layer -- statistics per group, running statistics advanced once per group with its momentum"""
        assert dil == 1 or not transposed
        # reflect = r > 0: the block stands for ReflectionPad2d(r) + this conv (whose own padding is 0).  Where the LDS-DMA kernels
        # cover the shape the pad is folded into their loaders, elsewhere it is an explicit pass (decided per call in _ConvBlock)
        assert not reflect or (pad == 0 and not transposed and stride == 1 and dil == 1)
        self.reflect = int(reflect)
        # passthrough: the block also returns an alias of its input for a second consumer (a residual junction); the alias'
        # cotangent is added to the input gradient inside the input-gradient kernel (rdfc_conv2d_dgrad_add)
        self.passthrough = bool(passthrough) and not transposed
        self.group_momenta = tuple(group_momenta) if group_momenta is not None and len(group_momenta) > 1 else None
        self.k, self.stride, self.pad, self.transposed, self.out_pad, self.dil = k, stride, pad, transposed, out_pad, dil
        self.norm, self.act, self.slope, self.eps, self.momentum = norm, act, slope, eps, momentum
        self.out_f32, self.wscale = out_f32, wscale


def _geometry(spec, x, weight):
    """-> (desc for conv-primitive orientation, Cout, Ho, Wo)"""
    N, H, W, Cin = x.shape
    k, s, p = spec.k, spec.stride, spec.pad + spec.reflect
    if not spec.transposed:
        Cout = weight.shape[0]
        Ho = (H + 2 * p - spec.dil * (k - 1) - 1) // s + 1
        Wo = (W + 2 * p - spec.dil * (k - 1) - 1) // s + 1
    else:   # ConvTranspose2d weight [Cin, Cout, k, k]
        Cout = weight.shape[1]
        Ho = (H - 1) * s - 2 * p + k + spec.out_pad
        Wo = (W - 1) * s - 2 * p + k + spec.out_pad
    return Cout, Ho, Wo


_BIAS_IN_APPLY = os.environ.get('RDFC_BIAS_IN_APPLY', '1') != '0'   # RDFC_BIAS_IN_APPLY=0: the bias gradient behind a norm layer by its own pass (A/B)
_STEM_FOLD = os.environ.get('RDFC_STEM_FOLD', '1') != '0'       # RDFC_STEM_FOLD=0: act_bwd + bias sum + weight gradient as three passes (A/B)


def _stem_backward(ctx, spec, x, weight, dz, yout, d_alias):
    """the backward of a conv + bias + ReLU / LeakyReLU block that needs NO input gradient and whose parameters' .grad buffers take
    accumulation (FlatAdam's): one launch for weight and bias gradient with the activation's backward folded in.  -> True if it ran"""
    if (not _STEM_FOLD or ctx.needs_input_grad[0] or d_alias is not None or spec.reflect or spec.transposed or spec.wscale != 1.0 or not x.is_cuda
            or spec.act not in ('relu', 'lrelu') or not ctx.needs_input_grad[1]):
        return False
    g, bg = weight.grad, (ctx.bias.grad if ctx.has_bias else None)
    if not (g is not None and g.dtype == torch.float32 and hip.is_ohwi(g) and not g.requires_grad):
        return False
    if ctx.has_bias and not (ctx.needs_input_grad[2] and _accumulable(bg)):
        return False
    if yout.dtype != x.dtype or dz.data_ptr() % 16 or yout.data_ptr() % 16 or hip.ld(yout) % 8:
        return False
    Cout = weight.shape[0]
    d = hip.conv_desc(x.shape, Cout, spec.k, spec.stride, spec.pad, spec.dil, ldx=hip.ld(x), ldy=hip.ld(dz), dtype=hip.dt(x), act=spec.act,
                      slope=spec.slope)
    if not hip.conv_wgrad_act_supported(d):
        return False
    if _WGRAD_STREAM and _multi_stream:
        stream, keep = _side_stream(x.device)
        stream.wait_stream(torch.cuda.current_stream(x.device))
        if hip.PROFILE is None:
            hip.conv_wgrad_act(d, x, dz, yout, g, bg, stream=stream.cuda_stream)
        else:
            with torch.cuda.stream(stream):                                 # (per-kernel events record on the current stream)
                hip.conv_wgrad_act(d, x, dz, yout, g, bg)
        keep.append((x, dz, yout))          # their memory must not be recycled before the join
    else:
        hip.conv_wgrad_act(d, x, dz, yout, g, bg)
    return True


class _ConvBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, res, prelu, spec, training, run_mean, run_var, cat_with=None, chain=None):
        dtype = x.dtype
        ctx.chain, ctx.cat_chain = chain if chain is not None else (None, None)
        N, H, W, Cin = x.shape
        Cout, Ho, Wo = _geometry(spec, x, weight)
        wf, wb = _packed(weight, dtype, spec.wscale)
        has_norm = spec.norm != 'none'
        epi_act = 'none' if has_norm else spec.act
        out_f32 = spec.out_f32 and not has_norm
        y = torch.empty(N, Ho, Wo, Cout, dtype=torch.float32 if out_f32 else dtype, device=x.device)
        fused = None                # (mean, rstd) when the conv's epilogue produced the normalisation statistics
        live_stats = has_norm and not (spec.norm == 'bn' and not training)
        gm = spec.group_momenta if spec.norm == 'bn' and training else None
        if gm is not None:
            assert N % len(gm) == 0, f'batch {N} does not split into {len(gm)} stacked batches'
        folded = False
        if spec.reflect:
            # ReflectionPad2d(r) + conv: folded into the conv's loader where the LDS-DMA kernels cover the shape (forward and
            # weight gradient), else an explicit pad pass in front of the pad-0 conv
            d = hip.conv_desc(x.shape, Cout, spec.k, 1, spec.reflect, 1, ldx=hip.ld(x), ldy=Cout, dtype=hip.dt(x),
                              act=epi_act, slope=spec.slope, out_f32=out_f32)
            folded = x.is_cuda and hip.conv_reflect_supported(d)
            x_alias = x
            if folded:
                hip.conv_forward(d, x, wf, bias, y, reflect=True)
            else:
                x = hip.reflect_pad(x.contiguous(), spec.reflect)          # (saved for the weight gradient in place of x)
                d = hip.conv_desc(x.shape, Cout, spec.k, 1, 0, 1, ldx=hip.ld(x), ldy=Cout, dtype=hip.dt(x),
                                  act=epi_act, slope=spec.slope, out_f32=out_f32)
                hip.conv_forward(d, x, wf, bias, y)
        elif not spec.transposed:
            d = hip.conv_desc(x.shape, Cout, spec.k, spec.stride, spec.pad, spec.dil, ldx=hip.ld(x), ldy=Cout, dtype=hip.dt(x),
                              act=epi_act, slope=spec.slope, out_f32=out_f32)
            # groups of the statistics: per image (instance norm), per stacked batch, or the whole batch
            G_ = N if spec.norm == 'in' else (len(gm) if gm is not None else 1)
            if live_stats and hip.conv_stats_nchunk(d, G_) > 0:
                fused = hip.conv_forward_partials(d, x, wf, bias, y, G_)     # the conv's epilogue wrote the partial statistics rows
            else:
                hip.conv_forward(d, x, wf, bias, y)
        else:   # equivalent conv maps the (larger) output map onto x
            d = hip.conv_desc((N, Ho, Wo, Cout), Cin, spec.k, spec.stride, spec.pad, ldx=Cout, ldy=hip.ld(x),
                              dtype=hip.dt(x), act=epi_act, slope=spec.slope, out_f32=out_f32, out_hw=(H, W))
            hip.conv_dgrad(d, x, wb, bias, y)
        ctx.spec, ctx.training, ctx.shapes = spec, training, (Cout, Ho, Wo)
        ctx.folded, ctx.in_shape = folded, (N, H, W, Cin)
        if not spec.reflect:
            x_alias = x
        ctx.has_bias = bias is not None
        ctx.bias = bias                    # (a parameter: kept for its .grad, into which the bias gradient accumulates directly)
        ctx.cat = False
        if not has_norm:
            assert cat_with is None, 'cat_with: blocks with a normalisation layer only'
            ctx.save_for_backward(x, weight, y if spec.act != 'none' else None)
            return (y, x_alias.view(x_alias.shape)) if spec.passthrough else y
        G = N if spec.norm == 'in' else (len(gm) if gm is not None else 1)
        if spec.norm == 'bn' and not training:
            mean = run_mean.view(1, -1).float()
            rstd = torch.rsqrt(run_var.view(1, -1).float() + spec.eps)
            z, buf = _out_for_cat(N, Ho, Wo, Cout, dtype, x.device, cat_with)
            hip.norm_apply(y, mean, rstd, gamma, beta, res, z, spec.act, spec.slope, prelu)
        else:
            upd = spec.norm == 'bn' and run_mean is not None
            z, buf = _out_for_cat(N, Ho, Wo, Cout, dtype, x.device, cat_with)
            if fused is not None:
                mom = gm if gm is not None else ((spec.momentum,) if upd else None)
                mean, rstd = hip.norm_from_partials(fused[0], fused[1], y, G, spec.eps, gamma, beta, res, z, spec.act, spec.slope, prelu,
                                                    run_mean if upd else None, run_var if upd else None, mom)
            else:
                mean, rstd = hip.norm_forward(y, G, spec.eps, gamma, beta, res, z, spec.act, spec.slope, prelu,
                                              run_mean if upd else None, run_var if upd else None, spec.momentum, group_momenta=gm)
        ctx.save_for_backward(x, weight, y, mean, rstd, gamma, beta, res, prelu)
        if cat_with is not None:            # [block output | cat_with]: the block wrote its channels of the concatenation itself
            assert not spec.passthrough
            hip.copy_window(cat_with, buf[..., Cout:])
            ctx.cat = True
            return buf
        return (z, x_alias.view(x_alias.shape)) if spec.passthrough else z

    @staticmethod
    def backward(ctx, dz, d_alias=None):
        spec = ctx.spec
        Cout, Ho, Wo = ctx.shapes
        has_norm = spec.norm != 'none'
        g_gamma = g_beta = g_res = g_prelu = g_bias = g_cat = None
        if ctx.cat:                          # cotangent of [block output | cat_with]: two channel slices, read in place
            g_cat = dz[..., Cout:]
            dz = dz[..., :Cout]
            if ctx.cat_chain is not None and ctx.needs_input_grad[11]:
                g_cat = ctx.cat_chain.deposit(g_cat)      # a conv consumer of the same activation adds it in its epilogue (_Chain)
        if has_norm:
            x, weight, y, mean, rstd, gamma, beta, res, prelu = ctx.saved_tensors
            dz = dz if dz.dtype == x.dtype else dz.to(x.dtype)
            frozen = spec.norm == 'bn' and not ctx.training
            # affine gradients straight into persistent .grad buffers (FlatAdam's) when there is one statistics group
            direct = (gamma is not None and beta is not None and ctx.needs_input_grad[3]
                      and ctx.needs_input_grad[4] and _accumulable(gamma.grad) and _accumulable(beta.grad))
            # a conv bias in front of the norm layer (ResnetGenerator): its gradient = the sum of dy over the pixels, taken by the apply pass
            # while it stores dy (rdfc_norm_backward_dxsum) instead of by one more pass over dy
            bias_in_apply = (_BIAS_IN_APPLY and ctx.has_bias and ctx.needs_input_grad[2] and _accumulable(ctx.bias.grad) and y.is_cuda
                             and y.dtype != torch.float64)
            dy, g_res, sums = hip.norm_backward(dz, y, mean, rstd, gamma, beta, res, spec.act, spec.slope, prelu,
                              frozen=frozen, want_dres=res is not None and ctx.needs_input_grad[5],
                              acc_beta=beta.grad if direct else None, acc_gamma=gamma.grad if direct else None,
                              acc_bias=ctx.bias.grad if bias_in_apply else None)
            if gamma is not None and not direct:
                g_gamma, g_beta = (sums[0, 1], sums[0, 0]) if sums.shape[0] == 1 else (sums[:, 1].sum(0), sums[:, 0].sum(0))
            if prelu is not None:
                g_prelu = sums[:, 2].sum().view(1)
        else:
            x, weight, yout = ctx.saved_tensors
            # a channel slice of a concatenation's cotangent is read in place through its pixel stride -- when base pointer AND pixel
            # stride keep every pixel's vector aligned (a slice of a 64 + 3 channel map does not: copy it)
            if not (hip.is_pm(dz) and dz.data_ptr() % 16 == 0 and hip.ld(dz) % (16 // dz.element_size()) == 0):
                dz = dz.contiguous()
            if spec.act != 'none':
                if dz.dtype != x.dtype:
                    dz = dz.to(x.dtype)
                if _stem_backward(ctx, spec, x, weight, dz, yout, d_alias):
                    # the stems (no input gradient, thin input): ReLU's backward, the bias gradient and the weight gradient in ONE pass over
                    # dz and y (rdfc_conv2d_wgrad_act) instead of three passes and a dy map in between
                    return None, None, None, g_gamma, g_beta, g_res, g_prelu, None, None, None, None, g_cat, None
                dy = hip.act_bwd(dz, yout, spec.act, spec.slope)
            else:
                dy = dz if dz.dtype == x.dtype else dz.to(x.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2] and not (has_norm and bias_in_apply):
            bg = ctx.bias.grad
            if _accumulable(bg) and not dy.dtype == torch.float64:
                hip.channel_sum(dy, acc=bg)         # straight into the persistent .grad (FlatAdam's)
            else:
                g_bias = hip.channel_sum(dy)
        N, H, W, Cin = x.shape
        wf, wb = _packed(weight, x.dtype, spec.wscale)
        g_x = g_w = None
        if spec.reflect:
            r = spec.reflect
            N, H, W, Cin = ctx.in_shape                       # the unpadded input (x itself when the pad was folded, else x is padded)
            if ctx.needs_input_grad[0]:
                # input gradient on the padded geometry (a pad-0 conv's), folded back by the pad's adjoint (+ the alias' cotangent)
                d = hip.conv_desc((N, H + 2 * r, W + 2 * r, Cin), Cout, spec.k, 1, 0, 1, ldx=Cin, ldy=hip.ld(dy), dtype=hip.dt(x))
                g_pad = torch.empty(N, H + 2 * r, W + 2 * r, Cin, dtype=x.dtype, device=x.device)
                hip.conv_dgrad(d, dy, wb, None, g_pad)
                if d_alias is not None and d_alias.dtype != x.dtype:
                    d_alias = d_alias.to(x.dtype)
                g_x = hip.reflect_pad_adjoint(g_pad, r, addend=d_alias)
            elif d_alias is not None:
                g_x = d_alias
            if ctx.needs_input_grad[1]:
                if ctx.folded:
                    d = hip.conv_desc(x.shape, Cout, spec.k, 1, r, 1, ldx=hip.ld(x), ldy=hip.ld(dy), dtype=hip.dt(x))
                else:
                    d = hip.conv_desc(x.shape, Cout, spec.k, 1, 0, 1, ldx=hip.ld(x), ldy=hip.ld(dy), dtype=hip.dt(x))
                g_w = _wgrad(d, x, dy, weight, Cout, Cin, spec.k, spec.wscale, reflect=ctx.folded)
        elif not spec.transposed:
            if ctx.needs_input_grad[0]:
                d = hip.conv_desc(x.shape, Cout, spec.k, spec.stride, spec.pad, spec.dil, ldx=Cin, ldy=hip.ld(dy), dtype=hip.dt(x))
                g_x = torch.empty(N, H, W, Cin, dtype=x.dtype, device=x.device)
                if d_alias is not None and d_alias.dtype != x.dtype:
                    d_alias = d_alias.to(x.dtype)
                chain = ctx.chain
                if chain is not None and (spec.passthrough or d_alias is not None):
                    # the block also owns an alias of x (residual / downsample path): its result IS the activation's cotangent as far as
                    # this consumer goes -- take what the chain holds as a second addend, hand the sum to autograd
                    hip.conv_dgrad(d, dy, wb, None, g_x, addend=d_alias, addend2=chain.take(x.dtype))
                else:
                    hip.conv_dgrad(d, dy, wb, None, g_x, addend=d_alias if chain is None else chain.take(x.dtype))
                    if chain is not None:
                        g_x = chain.put(g_x)
            elif d_alias is not None:
                g_x = d_alias
            if ctx.needs_input_grad[1]:
                d = hip.conv_desc(x.shape, Cout, spec.k, spec.stride, spec.pad, spec.dil, ldx=hip.ld(x), ldy=hip.ld(dy), dtype=hip.dt(x))
                g_w = _wgrad(d, x, dy, weight, Cout, Cin, spec.k, spec.wscale)
        else:
            if ctx.needs_input_grad[0]:
                d = hip.conv_desc((N, Ho, Wo, Cout), Cin, spec.k, spec.stride, spec.pad, ldx=hip.ld(dy), ldy=Cin,
                                  dtype=hip.dt(x), out_hw=(H, W))
                g_x = torch.empty(N, H, W, Cin, dtype=x.dtype, device=x.device)
                hip.conv_forward(d, dy, wf, None, g_x)
            if ctx.needs_input_grad[1]:
                d = hip.conv_desc((N, Ho, Wo, Cout), Cin, spec.k, spec.stride, spec.pad, ldx=hip.ld(dy), ldy=hip.ld(x),
                                  dtype=hip.dt(x), out_hw=(H, W))
                g_w = _wgrad(d, dy, x, weight, Cin, Cout, spec.k, spec.wscale)     # [O = Cin][T][I = Cout]
        return g_x, g_w, g_bias, g_gamma, g_beta, g_res, g_prelu, None, None, None, None, g_cat, None


def conv_block(x, weight, bias=None, gamma=None, beta=None, res=None, prelu=None, *, spec, training=True,
               run_mean=None, run_var=None, cat_with=None):
    """cat_with (blocks with a normalisation layer): return cat((block output, cat_with), channels) -- the normalise pass writes the
    block's channels of the concatenation directly (no separate copy of them); same shape [N, Ho, Wo, *] required"""
    chain = getattr(x, '_rdfc_chain', None)
    if chain is not None and (spec.transposed or spec.reflect):
        chain = None                         # (those input-gradient paths take no addend: the sum falls back to _FanOut.backward)
    cat_chain = getattr(cat_with, '_rdfc_chain', None) if cat_with is not None else None
    return _ConvBlock.apply(x, weight, bias, gamma, beta, res, prelu, spec, training, run_mean, run_var, cat_with, (chain, cat_chain))


def _out_for_cat(N, Ho, Wo, Cout, dtype, device, cat_with):
    """-> (where the block writes its output, the concatenation buffer or None)"""
    if cat_with is None:
        return torch.empty(N, Ho, Wo, Cout, dtype=dtype, device=device), None
    assert cat_with.shape[:3] == (N, Ho, Wo) and cat_with.dtype == dtype, (tuple(cat_with.shape), (N, Ho, Wo), cat_with.dtype, dtype)
    buf = torch.empty(N, Ho, Wo, Cout + cat_with.shape[3], dtype=dtype, device=device)
    return buf[..., :Cout], buf


class _WAdaIN(torch.autograd.Function):
    """model_utils.py:72-90 with EqualLR's sqrt(2/fan_in) folded into the packed 1x1 weights."""

    @staticmethod
    def forward(ctx, x, style, weight_orig, bias):
        N, H, W, C = x.shape
        Cs = style.shape[3]
        scale = math.sqrt(2.0 / Cs)
        w4 = weight_orig.view(2 * C, Cs, 1, 1)
        wf, _ = _packed(w4, x.dtype, scale)
        gb = torch.empty(N, H, W, 2 * C, dtype=x.dtype, device=x.device)
        d = hip.conv_desc(style.shape, 2 * C, 1, 1, 0, ldx=hip.ld(style), ldy=2 * C, dtype=hip.dt(x))
        hip.conv_forward(d, style, wf, bias, gb)
        mean, rstd = hip.norm_stats(x, N, 1e-5)
        out = torch.empty(N, H, W, C, dtype=x.dtype, device=x.device)
        hip.adain_apply(x, mean, rstd, gb, out)
        ctx.save_for_backward(x, style, weight_orig, gb, mean, rstd)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        x, style, weight_orig, gb, mean, rstd = ctx.saved_tensors
        N, H, W, C = x.shape
        Cs = style.shape[3]
        dx, dgb = hip.adain_backward(dout, x, mean, rstd, gb)
        w4 = weight_orig.view(2 * C, Cs, 1, 1)
        _, wb = _packed(w4, x.dtype, ctx.scale)
        d = hip.conv_desc(style.shape, 2 * C, 1, 1, 0, ldx=Cs, ldy=2 * C, dtype=hip.dt(x))
        dstyle = torch.empty(N, H, W, Cs, dtype=x.dtype, device=x.device)
        hip.conv_dgrad(d, dgb, wb, None, dstyle)
        d = hip.conv_desc(style.shape, 2 * C, 1, 1, 0, ldx=hip.ld(style), ldy=2 * C, dtype=hip.dt(x))
        dw = hip.conv_wgrad(d, style, dgb)
        g_w = hip.unpack_wgrad(dw, 2 * C, Cs, 1, 1, ctx.scale).view(2 * C, Cs)
        return dx, dstyle, g_w, hip.channel_sum(dgb)


def wadain(x, style, weight_orig, bias):
    return _WAdaIN.apply(x, style, weight_orig, bias)


class _CropCat(torch.autograd.Function):
    """rdf_generator.py:244-260: fd loses trailing rows/cols beyond fe's size, then cat((fd, fe), C)."""

    @staticmethod
    def forward(ctx, fd, fe, chain=None):
        ctx.chain = chain
        N, He, We, Ce = fe.shape
        Cd = fd.shape[3]
        out = torch.empty(N, He, We, Cd + Ce, dtype=fd.dtype, device=fd.device)
        hip.copy_window(fd, out[..., :Cd])
        hip.copy_window(fe, out[..., Cd:])
        ctx.fd_shape = fd.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Hd, Wd, Cd = ctx.fd_shape
        _, He, We, _ = dout.shape
        g_fe = dout[..., Cd:]                       # channel slice: consumers take ld > C
        if (Hd, Wd) == (He, We):
            g_fd = dout[..., :Cd]
        else:
            g_fd = hip.zero_(torch.empty(N, Hd, Wd, Cd, dtype=dout.dtype, device=dout.device))
            hip.copy_window(dout[..., :Cd], g_fd)
        if ctx.chain is not None and ctx.needs_input_grad[1]:
            g_fe = ctx.chain.deposit(g_fe)          # (a chained fan-out's skip connection: functional._Chain)
        return g_fd, g_fe, None


def crop_cat(fd, fe):
    return _CropCat.apply(fd, fe, getattr(fe, '_rdfc_chain', None))


class _Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        N, H, W, Ca = a.shape
        out = torch.empty(N, H, W, Ca + b.shape[3], dtype=a.dtype, device=a.device)
        hip.copy_window(a, out[..., :Ca])
        hip.copy_window(b, out[..., Ca:])
        ctx.ca = Ca
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout[..., :ctx.ca], dout[..., ctx.ca:]


def cat(a, b):
    return _Cat.apply(a, b)


class _Chain:
    """Running sum of the cotangents of one activation (fanout(..., chained=True)), kept out of autograd's hands so that no rdfc_sum_maps
    pass is needed.  Three kinds of consumer:
      * a conv block reading the alias (put): its input-gradient launch adds the sum so far in its epilogue (rdfc_conv2d_dgrad_add) and
        leaves the result here instead of returning it; the last of the n hands the total to autograd;
      * a concatenation's second operand (deposit): the channel slice of the concatenation's cotangent is left here for a conv consumer;
      * a conv block that also owns an alias of its input (passthrough: the next encoder stage's first block): takes what is here as a SECOND
        epilogue addend (rdfc_conv2d_dgrad_add2) and returns the sum to autograd itself.
    Whatever order the backward nodes run in, and whichever of them never run: what is still here when _FanOut.backward runs is one more
    cotangent for it to add."""
    __slots__ = ('n', 'seen', 'partial', 'stream')

    def __init__(self, n):
        self.n, self.seen, self.partial, self.stream = n, 0, None, None

    def take(self, dtype):
        """-> the sum so far (None for the first consumer), safe to read on the current stream"""
        p = self.partial
        if p is None or p.dtype != dtype:
            return None
        cur = torch.cuda.current_stream(p.device)
        if self.stream is not None and self.stream != cur:
            cur.wait_stream(self.stream)
            p.record_stream(cur)
        self.partial = None
        return p

    def put(self, total):
        """total = this consumer's input gradient (+ what take() returned).  -> what the consumer returns to autograd"""
        self.seen += 1
        if self.partial is not None:                 # take() declined it (dtype): leave both to _FanOut.backward
            return total
        if self.seen < self.n:
            self.partial, self.stream = total, torch.cuda.current_stream(total.device)
            return None
        return total

    def deposit(self, g):
        """a consumer that cannot add (the channel slice of a concatenation's cotangent): leave g for a conv consumer to take.
        -> what the consumer returns to autograd (None: taken over by the chain)"""
        self.seen += 1
        if self.partial is not None or not g.is_cuda:
            return g
        self.partial, self.stream = g, torch.cuda.current_stream(g.device)
        return None

    def drain(self):
        p, self.partial, self.seen, self.stream = self.partial, None, 0, None
        return p


class _FanOut(torch.autograd.Function):
    """An activation with n consumers: n aliases forward, ONE rdfc_sum_maps pass backward (autograd would add the
    cotangents pairwise with stock elementwise kernels, after copying channel-slice cotangents contiguous)."""

    @staticmethod
    def forward(ctx, x, n, chain=None):
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        ctx.chain = chain
        ctx.set_materialize_grads(False)          # an alias without a cotangent stays None (not a map of zeros to add)
        return tuple(x.view(x.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        shape, dtype, dev = ctx.meta
        live = []
        if ctx.chain is not None:
            st = ctx.chain.stream
            p = ctx.chain.drain()
            if p is not None:
                cur = torch.cuda.current_stream(dev)
                if st is not None and st != cur:
                    cur.wait_stream(st)
                    p.record_stream(cur)
                gs = gs + (p,)
        for g in gs:
            if g is None:
                continue
            if g.dtype != dtype:
                g = g.to(dtype)
            if g.stride(3) != 1 and g.shape[3] != 1:
                g = g.contiguous()
            live.append(g)
        if not live:
            return None, None, None
        if len(live) == 1:
            return live[0], None, None
        out = None
        while len(live) > 4:
            out = hip.sum_maps(live[:4])
            live = [out] + live[4:]
        return hip.sum_maps(live), None, None




def fanout(x, n, chained=False):
    """x -> n aliases whose gradients are summed by one HIP pass; chained=True (every consumer is a conv_block): summed inside the
    consumers' input-gradient launches instead (_Chain)"""
    if n <= 1 or not x.requires_grad:
        return (x,) * n
    if chained and x.is_cuda:
        chain = _Chain(n)
        outs = _FanOut.apply(x, n, chain)
        for o in outs:
            o._rdfc_chain = chain
        return outs
    return _FanOut.apply(x, n)


class _FanOutFlat(torch.autograd.Function):
    """fanout for tensors of any layout-free shape (upstream's NCHW f32 images between the networks)"""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view(x.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        while len(live) > 1:
            live = [hip.sum_flat(live[:4])] + live[4:]
        return live[0], None


class _BackwardMarker(torch.autograd.Function):
    """identity whose backward calls `fn()` first: placed on a network's input, it fires when that network's whole backward
    has been enqueued (data-parallel bucket exchange, rdfc_gan.py)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def backward_marker(x, fn):
    return _BackwardMarker.apply(x, fn) if x.requires_grad else x


def fanout_flat(x, n):
    if n <= 1 or not x.requires_grad or not x.is_cuda:
        return (x,) * n
    return _FanOutFlat.apply(x.contiguous(), n)


class _CombineLosses(torch.autograd.Function):
    """total = post * sum_i w_i * l_i (sequential f32 adds, in order) plus the scaled terms, one launch each way"""

    @staticmethod
    def forward(ctx, weights, post, *terms):
        out = hip.combine_scalars(terms, weights, post)
        ctx.meta = (tuple(weights), post, terms[0].device)
        total, scaled = out[len(terms)], out[:len(terms)]
        ctx.mark_non_differentiable(scaled)           # logged values only
        return total, scaled

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        weights, post, dev = ctx.meta
        g = hip.combine_scalars_backward(weights, post, g_total, dev)
        return (None, None) + tuple(g[i] for i in range(len(weights)))


def combine_losses(terms, post=1.0):
    """terms: [(0-dim loss tensor, python weight), ...] -> (total, f32[n] of the weighted terms, detached, for logging)"""
    ts = [t for t, _ in terms]
    ws = [float(w) for _, w in terms]
    if not ts[0].is_cuda:
        scaled = [t * w for t, w in zip(ts, ws)]
        total = scaled[0]
        for v in scaled[1:]:
            total = total + v
        return total * post if post != 1.0 else total, torch.stack([v.detach() for v in scaled])
    return _CombineLosses.apply(ws, post, *ts)


class _ReflectPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad, passthrough):
        ctx.pad = pad
        out = hip.reflect_pad(x.contiguous(), pad)
        return (out, x.view(x.shape)) if passthrough else out

    @staticmethod
    def backward(ctx, dy, d_alias=None):
        if d_alias is not None and d_alias.dtype != dy.dtype:
            d_alias = d_alias.to(dy.dtype)
        return hip.reflect_pad_adjoint(dy.contiguous(), ctx.pad, addend=d_alias), None, None


def reflect_pad(x, pad, passthrough=False):
    """passthrough: -> (padded, alias of x) for a second consumer of x; the alias' cotangent is added inside the adjoint kernel"""
    return _ReflectPad.apply(x, pad, passthrough)


class _ToPM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        return hip.nchw_to_pm(x.float(), dtype)

    @staticmethod
    def backward(ctx, g):
        return hip.pm_to_nchw(g), None


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return hip.pm_to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return hip.nchw_to_pm(g.contiguous(), ctx.dtype)


def to_pm(x_nchw, dtype=None):
    return _ToPM.apply(x_nchw, dtype or _compute_dtype)


def to_nchw(x_pm):
    """[N,H,W,C] -> [N,C,H,W] f32.  1-channel f32 maps are reinterpreted, not copied."""
    if x_pm.shape[3] == 1 and x_pm.dtype == torch.float32 and x_pm.is_contiguous():
        return x_pm.permute(0, 3, 1, 2)
    return _ToNCHW.apply(x_pm)


class _FusionHead(torch.autograd.Function):
    """rdf_generator.py:400-406 on 1-channel f32 maps: clamp, softmax(2), weighted sum."""

    @staticmethod
    def forward(ctx, d1, c1, d2, c2):
        d1, c1, d2, c2 = (t.contiguous() for t in (d1, c1, d2, c2))
        d2c, pred = hip.fusion_head(d1, c1, d2, c2)
        ctx.save_for_backward(d1, c1, d2, c2)
        return d2c, pred

    @staticmethod
    def backward(ctx, g_d2c, g_pred):
        d1, c1, d2, c2 = ctx.saved_tensors
        g_d2c = g_d2c.contiguous() if g_d2c is not None else None
        g_pred = g_pred.contiguous() if g_pred is not None else None
        return tuple(hip.fusion_head_bwd(d1, c1, d2, c2, g_pred, g_d2c))


def fusion_head(d1, c1, d2, c2):
    return _FusionHead.apply(d1, c1, d2, c2)


class _WeightedLoss(torch.autograd.Function):
    """gan_loss.py L1_loss / L2_loss / mse_loss: sum_i w_i |p_i - t_i|^k, fused value + gradient."""

    @staticmethod
    def forward(ctx, pred, target, weight, kind):
        val, grad = hip.weighted_loss(pred, target, weight, kind, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return hip.scale_dev(grad, g), None, None, None


def weighted_loss(pred, target, weight, kind):
    return _WeightedLoss.apply(pred, target, weight, kind)


# ---- label net operators: labels_model.py / backbone_resnet50.py ----------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        y, idx = hip.maxpool2d(x, k, s, p)
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape), k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, k, s, p = ctx.geom
        return hip.maxpool2d_backward(dy.contiguous(), ctx.saved_tensors[0], shape, k, s, p), None, None, None


def max_pool2d(x, k, s, p):
    return _MaxPool.apply(x, k, s, p)


class _AdaptiveAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        ctx.shape = tuple(x.shape)
        return hip.adaptive_avgpool(x, Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        return hip.adaptive_avgpool_backward(dy.contiguous(), ctx.shape), None, None


def adaptive_avg_pool2d(x, size):
    Ho, Wo = (size, size) if isinstance(size, int) else size
    return _AdaptiveAvgPool.apply(x, Ho, Wo)


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, out_f32):
        ctx.meta = (tuple(x.shape), x.dtype)
        return hip.bilinear(x, Ho, Wo, out_f32=out_f32)

    @staticmethod
    def backward(ctx, dy):
        shape, dtype = ctx.meta
        if dy.dtype != torch.float32 and dy.dtype != dtype:
            dy = dy.to(dtype)
        return hip.bilinear_backward(dy.contiguous(), shape, dtype), None, None, None


def interpolate_bilinear(x, size, out_f32=False):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) on a pixel-major map"""
    return _Bilinear.apply(x, size[0], size[1], out_f32)


class _ScaleNC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return hip.scale_nc(x, scale)

    @staticmethod
    def backward(ctx, dy):
        return hip.scale_nc(dy.contiguous(), ctx.saved_tensors[0]), None


def dropout2d(x, p, training):
    """nn.Dropout2d: whole channels of a sample are zeroed with probability p, the rest scaled by 1 / (1 - p).  The keep
    mask comes from torch's generator on the device (one draw of N x C values, like upstream's feature_dropout)."""
    if not training or p == 0.0:
        return x
    N, _, _, C = x.shape
    keep = torch.empty(N, C, dtype=torch.float32, device=x.device).bernoulli_(1.0 - p).div_(1.0 - p)
    return _ScaleNC.apply(x, keep)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        loss, count = hip.cross_entropy(logits, labels, ignore_index)
        ctx.save_for_backward(logits, labels, count)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, count = ctx.saved_tensors
        return hip.cross_entropy_backward(logits, labels, count, g, ctx.ignore_index), None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """torch.nn.CrossEntropyLoss()(logits, labels) with pixel-major logits [N,H,W,C]"""
    return _CrossEntropy.apply(logits, labels, ignore_index)


def cross_entropy_nchw(logits_nchw, labels, ignore_index=-100):
    """the label net hands out upstream's [N, classes, H, W] shape as a view of its pixel-major map: undo the view (no copy)"""
    z = logits_nchw.permute(0, 2, 3, 1)
    return cross_entropy(z if z.is_contiguous() else z.contiguous(), labels, ignore_index)


class _NormalLosses(torch.autograd.Function):
    """nor_loss + manhattan_loss (lib/losses/gan_loss.py:122-132, 50-99) -> [nor, floor, wall, ceiling]; differentiable in `normal`"""

    @staticmethod
    def forward(ctx, normal, gt_normal, norm_mask, logits_nchw, labels3):
        out8 = hip.normal_losses(normal, gt_normal, norm_mask, logits_nchw, *labels3, with_counts=True)
        ctx.save_for_backward(normal, gt_normal, norm_mask, logits_nchw, out8)
        ctx.labels3 = labels3
        return out8[:4].clone()

    @staticmethod
    def backward(ctx, g4):
        normal, gt_normal, norm_mask, logits_nchw, out8 = ctx.saved_tensors
        return hip.normal_losses_backward(normal, gt_normal, norm_mask, logits_nchw, *ctx.labels3, out8, g4), None, None, None, None


def normal_losses(normal, gt_normal, norm_mask, logits_nchw, label_wall, label_floor, label_ceiling):
    return _NormalLosses.apply(normal, gt_normal, norm_mask, logits_nchw.detach() if logits_nchw is not None else None,
                               (label_wall, label_floor, label_ceiling))
