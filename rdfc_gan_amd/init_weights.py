"""Start-of-training parameter values, as RDFC-GAN/lib/models/init_weights.py:5-33 leaves them when RDFCGAN calls it
(rdfc_gan.py:120, always with the defaults init_type='normal', init_gain=0.02):

  * every module whose class name contains 'Conv' or 'Linear' and that owns a `weight`: weight ~ N(0, gain), bias = 0;
  * BatchNorm2d: weight ~ N(1, gain), bias = 0;
  * everything else untouched -- in particular EqualLinear's `weight_orig` keeps its N(0, 1) draw (upstream's holder has
    no `weight` attribute before its first forward), InstanceNorm2d has no affine parameters, PReLU keeps 0.25.

Modules are visited in `net.modules()` order, the order `net.apply` uses upstream, so that a seeded run draws the same
numbers for the same parameters.  Only the 'normal' scheme is reachable from the train step and only it is offered.
"""
import torch

from . import functional as Fh


def _is_contraction(m):
    name = type(m).__name__
    return ('Conv' in name or 'Linear' in name) and getattr(m, 'weight', None) is not None


@torch.no_grad()
def init_weights(net, init_type='normal', init_gain=0.02, verbose=True):
    if init_type != 'normal':
        raise NotImplementedError(f'initialization method [{init_type}] is not implemented')
    if verbose:
        print('initialize network with %s' % init_type)
    # net.apply() is children-first; reproduce that order (the draws come from torch's global generator)
    order = []

    def visit(m):
        for c in m.children():
            visit(c)
        order.append(m)

    visit(net)
    for m in order:
        if _is_contraction(m):
            m.weight.normal_(0.0, init_gain)
            if getattr(m, 'bias', None) is not None:
                m.bias.zero_()
        elif 'BatchNorm2d' in type(m).__name__:
            m.weight.normal_(1.0, init_gain)
            m.bias.zero_()
    Fh.invalidate_packed_weights()
