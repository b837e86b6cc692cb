"""Oracle networks: plain torch.nn restatement of the three hot-path nets.

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32, CPU, no custom kernels.

State-dict keys are identical to the reference so that one set of weights can be
loaded into the reference (in the build container), this oracle and the HIP
product modules.

Reference:
  G_A2B  RDFC-GAN/lib/models/generator/rdf_generator/rdf_generator.py:32-414
         .../encoder_decoder/encoder_decoder.py:5-61, common.py:29-61
         .../model_utils.py:7-90 (EqualLR / W-AdaIN)
  G_B2A  RDFC-GAN/lib/models/generator/resnet_generator.py:6-100
  D      RDFC-GAN/lib/models/discriminator/patch_gan_discriminator.py:6-40
         RDFC-GAN/lib/models/module/conv_norm_act.py:10-125
  ResNet-18 BasicBlock stack: torchvision 0.7.0 resnet18 (un-vendored dependency,
  common.py:11-17); restated from its published architecture.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nlspn as _nlspn


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def instance_norm(x, eps=1e-5):
    """InstanceNorm2d (no affine, biased variance) written out with elementwise ops.

    Deliberately NOT F.instance_norm: the PyTorch 2.10 CPU kernel behind it returns wrong
    input gradients when the incoming cotangent is channels-last strided, which happens at
    batch size 1 behind a ConvTranspose2d (found while pinning this oracle: autograd result
    differs from the closed form by O(1); see DESIGN.md "Oracle").  Forward values are identical."""
    mu = x.mean((2, 3), keepdim=True)
    var = x.var((2, 3), unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


class _IN(nn.Module):
    """parameter-free stand-in for nn.InstanceNorm2d(c) (no state_dict entries either way)"""

    def __init__(self, c=None):
        super().__init__()

    def forward(self, x):
        return instance_norm(x)


def _cna(cin, cout, k, s=1, p=0, bn=True, act=True):
    """conv [+BN] [+LeakyReLU(0.2)]  (common.py:29-43); conv bias iff no BN."""
    mods = [nn.Conv2d(cin, cout, k, s, p, bias=not bn)]
    if bn:
        mods.append(nn.BatchNorm2d(cout))
    if act:
        mods.append(nn.LeakyReLU(0.2))
    return nn.Sequential(*mods)


def _tna(cin, cout):
    """ConvTranspose2d 3x3 s2 p1 op1 + BN + LeakyReLU(0.2)  (common.py:46-61)."""
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 3, 2, 1, 1, bias=False),
                         nn.BatchNorm2d(cout), nn.LeakyReLU(0.2))


class BasicBlock(nn.Module):
    """torchvision BasicBlock: conv-bn-relu-conv-bn (+identity/downsample) relu."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        sc = x if self.downsample is None else self.downsample(x)
        return F.relu(y + sc)


def _res_stage(cin, cout, stride):
    return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))


class EncDec(nn.Module):
    """One branch: resnet18 layer1..4 as en2..en5, en6, de5..de2
    (encoder_decoder.py:40-61)."""

    def __init__(self, enc=(64, 128, 256, 512, 512), dec=(256, 128, 64, 64)):
        super().__init__()
        self.en2 = _res_stage(64, enc[0], 1)
        self.en3 = _res_stage(enc[0], enc[1], 2)
        self.en4 = _res_stage(enc[1], enc[2], 2)
        self.en5 = _res_stage(enc[2], enc[3], 2)
        self.en6 = _cna(enc[3], enc[4], 3, 2, 1)
        self.de5 = _tna(enc[4], dec[0])
        self.de4 = _tna(dec[0] + enc[3], dec[1])
        self.de3 = _tna(dec[1] + enc[2], dec[2])
        self.de2 = _tna(dec[2] + enc[1], dec[3])


class _EqLinear(nn.Module):
    """nn.Linear whose weight is stored as `weight_orig` and used scaled by
    sqrt(2/fan_in) (model_utils.py:7-50)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(cout))          # registration order as upstream (bias, weight_orig)
        self.weight_orig = nn.Parameter(torch.randn(cout, cin))

    def forward(self, x):
        return F.linear(x, self.weight_orig * math.sqrt(2.0 / self.weight_orig.shape[1]), self.bias)


class _EqWrap(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = _EqLinear(cin, cout)


class WAdaIN(nn.Module):
    """model_utils.py:53-90: gamma,beta = Linear(style) per pixel;
    out = gamma * InstanceNorm(x) + beta."""

    def __init__(self, c, cs):
        super().__init__()
        self.style = _EqWrap(cs, 2 * c)
        with torch.no_grad():
            self.style.linear.bias[:c] = 1.0
        self.c = c

    def forward(self, x, style):
        gb = self.style.linear(style.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        gamma, beta = gb[:, :self.c], gb[:, self.c:]
        return gamma * instance_norm(x) + beta


def _crop_cat(fd, fe):
    """rdf_generator.py:244-260: drop trailing rows/cols of fd, then concat."""
    return torch.cat((fd[:, :, :fe.shape[2], :fe.shape[3]], fe), dim=1)


class _ConvSigmoid(nn.Sequential):
    def __init__(self, cin):
        super().__init__(nn.Conv2d(cin, 1, 3, 1, 1), nn.Sigmoid())


class RDFGenerator(nn.Module):
    """G_A2B (rdf_generator.py:32-414), shipped configuration: resnet18 encoders,
    concat skips, W-AdaIN fusion, BN."""

    def __init__(self, use_nlspn_refine=False, nlspn_configs=None, semantic_channels_in=3, **_unused):
        super().__init__()
        self.use_nlspn_refine = use_nlspn_refine
        self.rgb_branch_en1 = _cna(semantic_channels_in, 64, 3, 1, 1, bn=False)
        self.rgb_branch_encoder_decoder = EncDec()
        self.rgb_pred_dec1 = _cna(128, 64, 3, 1, 1)
        self.rgb_pred_dec0 = _cna(128, 1, 3, 1, 1, bn=False, act=False)
        self.rgb_conf_dec1 = _cna(128, 32, 3, 1, 1)
        self.rgb_conf_dec0 = _ConvSigmoid(96)
        self.depth_branch_en1_rgb = _cna(semantic_channels_in, 48, 3, 1, 1, bn=False)
        self.depth_branch_en1_depth = _cna(1, 16, 3, 1, 1, bn=False)
        self.depth_branch_encoder_decoder = EncDec()
        self.id_dec1 = _cna(128, 64, 3, 1, 1)
        self.id_dec0 = _cna(128, 1, 3, 1, 1, bn=False, act=False)
        if use_nlspn_refine:
            nn_ = nlspn_configs['prop_kernel'] ** 2 - 1
            self.gd_dec1 = _cna(128, 64, 3, 1, 1)
            self.gd_dec0 = _cna(128, nn_, 3, 1, 1, bn=False, act=False)
        self.cf_dec1 = _cna(128, 32, 3, 1, 1)
        self.cf_dec0 = _ConvSigmoid(96)
        if use_nlspn_refine:
            self.nlspn_refine_module = _nlspn.NLSPNRefine(**nlspn_configs)
        for i, (c, cs) in enumerate([(512, 512), (768, 768), (384, 384), (192, 192), (128, 128)], 1):
            setattr(self, f'fuse_layer{i}', WAdaIN(c, cs))   # fuse_layer5 exists but is never used

    def forward(self, rgb, depth, normal):
        R, D = self.rgb_branch_encoder_decoder, self.depth_branch_encoder_decoder
        r1 = self.rgb_branch_en1(normal)
        d1 = torch.cat([self.depth_branch_en1_rgb(normal), self.depth_branch_en1_depth(depth)], 1)
        r = [None, r1]
        d = [None, d1]
        for name in ('en2', 'en3', 'en4', 'en5', 'en6'):
            r.append(getattr(R, name)(r[-1]))
            d.append(getattr(D, name)(d[-1]))
        rf, df = r[6], d[6]
        for lvl, fuse in zip((5, 4, 3, 2), (self.fuse_layer1, self.fuse_layer2, self.fuse_layer3, self.fuse_layer4)):
            rf_new = _crop_cat(getattr(R, f'de{lvl}')(fuse(rf, df)), r[lvl])
            df = _crop_cat(getattr(D, f'de{lvl}')(df), d[lvl])
            rf = rf_new
        depth1 = torch.tanh(self.rgb_pred_dec0(_crop_cat(self.rgb_pred_dec1(rf), r1)))
        conf1 = self.rgb_conf_dec0(_crop_cat(self.rgb_conf_dec1(rf), r1))
        init = torch.tanh(self.id_dec0(_crop_cat(self.id_dec1(df), d1)))
        conf = self.cf_dec0(_crop_cat(self.cf_dec1(df), d1))
        if self.use_nlspn_refine:
            guide = self.gd_dec0(_crop_cat(self.gd_dec1(df), d1))
            depth2 = self.nlspn_refine_module(init, guide, conf)
        else:
            depth2 = init
        depth2 = depth2.clamp(-1, 1)
        score = torch.softmax(torch.cat([conf1, conf], 1), 1)
        pred = (torch.cat([depth1, depth2], 1) * score).sum(1, keepdim=True)
        return dict(depth_map_1=depth1, confidence_map_1=conf1, depth_map_2=depth2,
                    confidence_map_2=conf, pred_depth=pred)


class _ResBlk(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), _IN(c), nn.LeakyReLU(0.01),
            nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), _IN(c))

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetGenerator(nn.Module):
    """G_B2A (resnet_generator.py:6-57) with norm_layer='IN2d' (convs have bias)."""

    def __init__(self, input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d', n_blocks=6, **_unused):
        super().__init__()
        assert norm_layer.lower() == 'in2d', 'oracle covers the shipped IN2d configuration'
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_channels, ngf, 7), _IN(ngf), nn.LeakyReLU(0.01)]
        c = ngf
        for _ in range(2):
            m += [nn.Conv2d(c, 2 * c, 3, 2, 1), _IN(2 * c), nn.PReLU(init=0.25)]
            c *= 2
        m += [_ResBlk(c) for _ in range(n_blocks)]
        for _ in range(2):
            m += [nn.ConvTranspose2d(c, c // 2, 3, 2, 1, 1), _IN(c // 2), nn.PReLU(init=0.25)]
            c //= 2
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_channels, 7), nn.Tanh()]
        self.model = nn.Sequential(*m)

    def forward(self, x):
        return self.model(x)


class _CM(nn.Module):
    """ConvModule (conv_norm_act.py:10-125) for the PatchGAN use: conv -> bn2d -> act."""

    def __init__(self, cin, cout, k, s, p, norm, act_slope):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, p, bias=not norm)
        if norm:
            self.bn2d = nn.BatchNorm2d(cout)
        self.has_norm, self.slope = norm, act_slope

    def forward(self, x):
        x = self.conv(x)
        if self.has_norm:
            x = self.bn2d(x)
        if self.slope is not None:
            x = F.leaky_relu(x, self.slope) if self.slope > 0 else F.relu(x)
        return x


class PatchGANDiscriminator(nn.Module):
    """patch_gan_discriminator.py:6-40; YAML: norm BN2d, activation LeakyReLU(0.2)."""

    def __init__(self, in_channels, out_channels=(64, 128, 256, 512, 1), kernel_size=(4,) * 5,
                 stride=(2, 2, 2, 1, 1), padding=(1,) * 5, activation='LeakyReLU', **_unused):
        super().__init__()
        ch = [in_channels] + list(out_channels)
        n = len(out_channels)
        slope = 0.2 if activation == 'LeakyReLU' else 0.0
        self.model = nn.Sequential(*[
            _CM(ch[i], ch[i + 1], kernel_size[i], stride[i], padding[i],
                norm=(0 < i < n - 1), act_slope=(None if i == n - 1 else slope)) for i in range(n)])

    def forward(self, x):
        return self.model(x)
