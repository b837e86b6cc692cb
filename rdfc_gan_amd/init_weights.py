"""init_weights with upstream's behaviour (RDFC-GAN/lib/models/init_weights.py:5-33):
every Conv*/Linear weight ~ N(0, gain), bias 0, BatchNorm2d weight ~ N(1, gain), bias 0.
EqualLinear's `weight_orig` is left at N(0,1) exactly as upstream leaves it (its `weight`
attribute does not exist before the first forward there)."""
from torch.nn import init

from . import functional as Fh


def init_weights(net, init_type='normal', init_gain=0.02, verbose=True):
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError(f'initialization method [{init_type}] is not implemented')
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    if verbose:
        print('initialize network with %s' % init_type)
    net.apply(init_func)
    Fh.invalidate_packed_weights()
