"""Parameter-holding building blocks with upstream-identical state_dict keys whose
forward runs the fused HIP blocks (rdfc_gan_amd.functional).

The torch.nn layers inside these containers (nn.Conv2d, nn.BatchNorm2d, ...) only own
parameters / buffers -- their own forward is never called.
"""
import contextlib

import os

import torch
import torch.nn as nn

from . import functional as Fh
from .functional import ConvSpec

# >1 while one forward stands in for several identical ones (rdfc_gan.py elides upstream's dead
# forwards): BatchNorm running statistics advance as if the forward had run BN_STAT_REPEAT times.
BN_STAT_REPEAT = 1
# Several calls of one network run as ONE pass over the concatenated batch (rdfc_gan.py's batched schedule): BN_GROUP_REPEATS
# = (r_0, ..., r_{G-1}) says that the batch is G stacked batches and that group g stands for r_g identical upstream forwards.
# BatchNorm then takes its statistics per group and advances the running statistics once per group, in order.
BN_GROUP_REPEATS = None

@contextlib.contextmanager
def bn_groups(repeats):
    """layers run inside see a batch made of len(repeats) stacked batches (see BN_GROUP_REPEATS)"""
    global BN_GROUP_REPEATS
    old, BN_GROUP_REPEATS = BN_GROUP_REPEATS, tuple(int(r) for r in repeats)
    try:
        yield
    finally:
        BN_GROUP_REPEATS = old


_ACTS = {nn.ReLU: ('relu', 0.0), nn.LeakyReLU: ('lrelu', None), nn.PReLU: ('prelu', 0.0), nn.Tanh: ('tanh', 0.0),
         nn.Sigmoid: ('sigmoid', 0.0)}


def _act_of(m):
    if m is None:
        return 'none', 0.0, None
    name, slope = _ACTS[type(m)]
    if name == 'lrelu':
        slope = m.negative_slope
    return name, slope, (m.weight if name == 'prelu' else None)


def _flush_batch_counter(module, prefix, keep_vars):
    pending = module.__dict__.get('_rdfc_pending_batches', 0)
    if pending:
        module.num_batches_tracked += pending
        module.__dict__['_rdfc_pending_batches'] = 0


def _count_batch(norm, k):
    """nn.BatchNorm2d.num_batches_tracked bookkeeping without a device op per layer and forward: counted on
    the host, written into the buffer whenever a state_dict is taken (the value is only ever read there;
    momentum is fixed, so it never enters the arithmetic)."""
    d = norm.__dict__
    if '_rdfc_pending_batches' not in d:
        norm.register_state_dict_pre_hook(_flush_batch_counter)
        # a loaded num_batches_tracked replaces the count: batches counted before the load must not be added to it
        norm.register_load_state_dict_post_hook(lambda module, incompatible: module.__dict__.__setitem__('_rdfc_pending_batches', 0))
        d['_rdfc_pending_batches'] = 0
    d['_rdfc_pending_batches'] += k


_spec_cache = {}


def run_conv(x, conv, norm=None, act=None, res=None, out_f32=False, passthrough=False, reflect=0, cat_with=None):
    """conv (nn.Conv2d | nn.ConvTranspose2d) [+ norm (BatchNorm2d | InstanceNorm2d)] [+ res] [+ act module]"""
    # the static description of the block is the same on every call of the same modules: cache it (a train step runs
    # ~180 blocks, each used to rebuild its ConvSpec from a dozen attribute lookups)
    key = (id(conv), id(norm), id(act), out_f32, BN_STAT_REPEAT, getattr(act, 'negative_slope', None), BN_GROUP_REPEATS, passthrough, reflect)
    hit = _spec_cache.get(key)
    if hit is not None and hit[0] is conv and hit[1] is norm and hit[2] is act:
        _c, _n, _a, spec, nname, prelu = hit
        gamma = beta = rm = rv = None
        training = True
        if nname == 'bn':
            gamma, beta, rm, rv = norm.weight, norm.bias, norm.running_mean, norm.running_var
            training = norm.training
            if training and norm.num_batches_tracked is not None:
                _count_batch(norm, sum(BN_GROUP_REPEATS) if BN_GROUP_REPEATS else BN_STAT_REPEAT)
        return Fh.conv_block(x, conv.weight, conv.bias, gamma, beta, res, prelu, spec=spec, training=training,
                             run_mean=rm, run_var=rv, cat_with=cat_with)
    transposed = isinstance(conv, nn.ConvTranspose2d)
    aname, slope, prelu = _act_of(act)
    if norm is None:
        nname = 'none'
    elif isinstance(norm, nn.BatchNorm2d):
        nname = 'bn'
    elif isinstance(norm, nn.InstanceNorm2d):
        nname = 'in'
    else:
        raise TypeError(f'unsupported norm layer {type(norm)}')
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    assert conv.kernel_size[0] == conv.kernel_size[1] and conv.dilation[0] == conv.dilation[1] and conv.groups == 1
    assert conv.padding[0] == conv.padding[1] and conv.stride[0] == conv.stride[1]
    momentum = getattr(norm, 'momentum', 0.1) or 0.1
    gmom = None
    if BN_GROUP_REPEATS and nname == 'bn':
        assert BN_STAT_REPEAT == 1
        gmom = tuple(1.0 - (1.0 - momentum) ** r for r in BN_GROUP_REPEATS)
        if len(gmom) == 1:
            momentum, gmom = gmom[0], None
    elif BN_STAT_REPEAT != 1:
        momentum = 1.0 - (1.0 - momentum) ** BN_STAT_REPEAT
    spec = ConvSpec(k, s, p, transposed, conv.output_padding[0] if transposed else 0, nname, aname, slope,
                    eps=getattr(norm, 'eps', 1e-5), momentum=momentum, out_f32=out_f32, dil=conv.dilation[0], group_momenta=gmom,
                    passthrough=passthrough, reflect=reflect)
    gamma = beta = rm = rv = None
    training = True
    if nname == 'bn':
        gamma, beta, rm, rv = norm.weight, norm.bias, norm.running_mean, norm.running_var
        training = norm.training
        if training and norm.num_batches_tracked is not None:
            _count_batch(norm, sum(BN_GROUP_REPEATS) if BN_GROUP_REPEATS else BN_STAT_REPEAT)
    _spec_cache[key] = (conv, norm, act, spec, nname, prelu)
    return Fh.conv_block(x, conv.weight, conv.bias, gamma, beta, res, prelu, spec=spec, training=training,
                         run_mean=rm, run_var=rv, cat_with=cat_with)


class ConvSeq(nn.Sequential):
    """nn.Sequential(conv, [norm], [act]) as built by upstream's conv_bn_relu / convt_bn_relu
    (encoder_decoder/common.py:29-61); forward = one fused block."""

    def forward(self, x, res=None, out_f32=False, cat_with=None):
        """cat_with: -> cat((block(x), cat_with), channels) with the block writing its part of the concatenation in place (blocks with a
        normalisation layer, equal map sizes)"""
        conv, norm, act = self[0], None, None
        for m in list(self)[1:]:
            if isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
                norm = m
            else:
                act = m
        return run_conv(x, conv, norm, act, res, out_f32, cat_with=cat_with)


def conv_bn_relu(cin, cout, kernel, stride=1, padding=0, bn=True, _in=False, relu=True):
    assert not (bn and _in)
    mods = [nn.Conv2d(cin, cout, kernel, stride, padding, bias=not bn)]
    if bn:
        mods.append(nn.BatchNorm2d(cout))
    if _in:
        mods.append(nn.InstanceNorm2d(cout))
    if relu:
        mods.append(nn.LeakyReLU(0.2, inplace=True))
    return ConvSeq(*mods)


def convt_bn_relu(cin, cout, kernel, stride=1, padding=0, output_padding=0, bn=True, relu=True):
    assert kernel % 2 == 1, f'only odd kernel is supported but kernel = {kernel}'
    mods = [nn.ConvTranspose2d(cin, cout, kernel, stride, padding, output_padding, bias=not bn)]
    if bn:
        mods.append(nn.BatchNorm2d(cout))
    if relu:
        mods.append(nn.LeakyReLU(0.2, inplace=True))
    return ConvSeq(*mods)


class BasicBlock(nn.Module):
    """torchvision resnet BasicBlock (keys conv1/bn1/conv2/bn2/downsample.{0,1})."""
    expansion = 1

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        # conv1 and the identity / downsample path share x: conv1's block hands out an alias whose cotangent its input-gradient
        # kernel adds in its epilogue (no separate summation pass)
        y, xr = run_conv(x, self.conv1, self.bn1, self.relu, passthrough=True)
        idt = xr if self.downsample is None else run_conv(xr, self.downsample[0], self.downsample[1], None)
        return run_conv(y, self.conv2, self.bn2, self.relu, res=idt)


def resnet_stage(cin, cout, stride, blocks):
    return nn.Sequential(*[BasicBlock(cin if i == 0 else cout, cout, stride if i == 0 else 1) for i in range(blocks)])
