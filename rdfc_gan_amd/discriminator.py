"""HIP-backed PatchGAN discriminator with upstream's signature and state_dict keys.

PatchGANDiscriminator <- RDFC-GAN/lib/models/discriminator/patch_gan_discriminator.py:6-40
ConvModule            <- RDFC-GAN/lib/models/module/conv_norm_act.py:10-125 (+ norm.py:17-64, activation.py:13-21)
build_discriminator   <- the git-ignored lib/models/discriminator/build_discriminator.py (call site rdfc_gan.py:83-84)
"""
import warnings

import torch.nn as nn

from . import functional as Fh
from .layers import run_conv

_NORMS = {'BN2d': ('bn2d', nn.BatchNorm2d), 'IN2d': ('in2d', nn.InstanceNorm2d)}


def build_norm_layer(cfg, num_features, postfix=''):
    assert isinstance(cfg, dict) and 'type' in cfg
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    cfg_.pop('in_discriminator', None)
    if layer_type not in _NORMS:
        raise KeyError(f'Unrecognized norm type{layer_type} (HIP path: {sorted(_NORMS)})')
    abbr, cls = _NORMS[layer_type]
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    layer = cls(num_features, affine=True, track_running_stats=True, **cfg_) if cls is nn.BatchNorm2d else cls(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


def act_layers(name):
    if name == 'LeakyReLU':
        return nn.LeakyReLU(negative_slope=0.2, inplace=True)
    if name == 'ReLU':
        return nn.ReLU(inplace=True)
    raise NotImplementedError(f'activation {name} is not on the HIP path (LeakyReLU / ReLU)')


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, activation='ReLU', inplace=True, order=('conv', 'norm', 'act')):
        super().__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        assert activation is None or isinstance(activation, str)
        if order != ('conv', 'norm', 'act') or dilation != 1 or groups != 1 or \
                (conv_cfg is not None and conv_cfg.get('type', 'Conv2d').lower() != 'conv2d'):
            raise NotImplementedError('HIP ConvModule covers Conv2d, order (conv, norm, act), dilation 1, groups 1')
        self.conv_cfg, self.norm_cfg, self.activation, self.inplace, self.order = conv_cfg, norm_cfg, activation, inplace, order
        self.with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = True if norm_cfg is None else 'bn' not in norm_cfg.get('type').lower()
        self.with_bias = bias
        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = self.conv.kernel_size, self.conv.stride, self.conv.padding
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        else:
            self.norm_name = None
        if self.activation:
            self.act = act_layers(self.activation)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out',
                                nonlinearity='leaky_relu' if self.activation == 'LeakyReLU' else 'relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    def forward(self, x, norm=True, out_f32=False):
        return run_conv(x, self.conv, self.norm if (norm and self.with_norm) else None,
                        self.act if self.activation else None, out_f32=out_f32)


class PatchGANDiscriminator(nn.Module):
    def __init__(self, in_channels, out_channels=(64, 128, 256, 512, 1), kernel_size=(4, 4, 4, 4, 4), stride=(2, 2, 2, 1, 1),
                 padding=(1, 1, 1, 1, 1), conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN2d'), activation='ReLU'):
        super().__init__()
        assert out_channels[-1] == 1, \
            f'The channel of the feature map obtained by the last conv layer must be 1, but got {out_channels[-1]}'
        ch = [in_channels] + list(out_channels)
        n = len(ch) - 1
        self.model = nn.Sequential(*[
            ConvModule(ch[i], ch[i + 1], kernel_size[i], stride[i], padding[i], conv_cfg=conv_cfg,
                       norm_cfg=None if i == 0 or i == n - 1 else norm_cfg, activation=None if i == n - 1 else activation)
            for i in range(n)])

    def forward(self, x):
        h = Fh.to_pm(x)
        mods = list(self.model)
        for i, m in enumerate(mods):
            h = m(h, out_f32=(i == len(mods) - 1))
        return Fh.to_nchw(h)


def build_discriminator(cfg_node):
    kwargs = dict(cfg_node)
    typ = kwargs.pop('type')
    if typ != 'PatchGANDiscriminator':
        raise KeyError(f'unknown discriminator type {typ}; available: PatchGANDiscriminator')
    return PatchGANDiscriminator(**kwargs)
