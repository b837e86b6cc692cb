"""G_label: PSPNet over a dilated ResNet-50 on the HIP kernels.

Same constructor, state_dict keys and outputs as upstream's
  PSPNet / _PSPModule     lib/models/generator/label_net/labels_model.py:12-94
  ResNet / Bottleneck     lib/models/generator/label_net/backbone_resnet50.py:76-235 (resnet50(): deep stem, dilated,
                          multi-grid, :263-278)
The torch.nn layers below only own parameters and buffers; the forward runs fused conv + BatchNorm + ReLU (+ residual)
blocks on pixel-major bf16 maps (layers.run_conv) and the label_ops.hip operators.  Logits come back as f32 with upstream's
[N, classes, H, W] shape (a channels-last view of the pixel-major map, which is what functional.cross_entropy consumes).
"""
import torch
import torch.nn as nn

from . import functional as Fh
from .layers import run_conv


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, dilation=1, project=False):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride, bias=False), nn.BatchNorm2d(4 * planes))

    def forward(self, x):
        y = run_conv(x, self.conv1, self.bn1, self.relu)
        y = run_conv(y, self.conv2, self.bn2, self.relu)
        idt = x if self.downsample is None else run_conv(x, self.downsample[0], self.downsample[1], None)
        return run_conv(y, self.conv3, self.bn3, self.relu, res=idt)


# (planes, stride of the first block, dilations): what resnet50()'s _make_layer calls produce (backbone_resnet50.py:176-221)
_STAGES = ((64, 1, (1, 1, 1)), (128, 2, (1, 1, 1, 1)), (256, 1, (1, 2, 2, 2, 2, 2)), (512, 1, (4, 8, 16)))


def _stage(cin, planes, stride, dilations):
    return nn.Sequential(*[Bottleneck(cin if i == 0 else 4 * planes, planes, stride if i == 0 else 1, d,
                                      project=(i == 0 and (stride != 1 or cin != 4 * planes)))
                           for i, d in enumerate(dilations)])


class _PSPModule(nn.Module):
    def __init__(self, in_channels, bin_sizes=(1, 2, 3, 6)):
        super().__init__()
        out_channels = in_channels // len(bin_sizes)
        self.bin_sizes = tuple(bin_sizes)
        self.stages = nn.ModuleList(nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                                                  nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)) for b in bin_sizes)
        self.bottleneck = nn.Sequential(nn.Conv2d(in_channels + out_channels * len(bin_sizes), out_channels, 1),
                                        nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True), nn.Dropout2d(0.1))

    def forward(self, f):
        size = f.shape[1:3]
        levels = [f]
        for b, st in zip(self.bin_sizes, self.stages):
            if self.training and f.shape[0] * b * b == 1:          # what nn.BatchNorm2d raises upstream at batch size 1
                raise ValueError(f'Expected more than 1 value per channel when training, got input size '
                                 f'torch.Size([1, {st[1].out_channels}, 1, 1])')
            levels.append(Fh.interpolate_bilinear(run_conv(Fh.adaptive_avg_pool2d(f, b), st[1], st[2], st[3]), size))
        bt = self.bottleneck
        y = run_conv(torch.cat(levels, 3), bt[0], bt[1], bt[2])
        return Fh.dropout2d(y, bt[3].p, self.training)


class PSPNet(nn.Module):
    def __init__(self, num_classes, in_channels=3, pretrained=False, use_aux=True):
        super().__init__()
        if pretrained:
            raise NotImplementedError('load a checkpoint with load_state_dict: the keys are upstream\'s')
        stem = nn.Sequential(nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                             nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                             nn.Conv2d(64, 128, 3, 1, 1, bias=False))
        self.initial = nn.Sequential(stem, nn.BatchNorm2d(128), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        cin = 128
        for i, (planes, stride, dil) in enumerate(_STAGES, 1):
            setattr(self, f'layer{i}', _stage(cin, planes, stride, dil))
            cin = 4 * planes
        self.use_aux = use_aux
        self.master_branch = nn.Sequential(_PSPModule(cin), nn.Conv2d(cin // 4, num_classes, 1))
        self.auxiliary_branch = nn.Sequential(nn.Conv2d(cin // 2, cin // 4, 3, padding=1, bias=False), nn.BatchNorm2d(cin // 4),
                                              nn.ReLU(inplace=True), nn.Dropout2d(0.1), nn.Conv2d(cin // 4, num_classes, 1))
        self._init_heads()

    def _init_heads(self):
        """initialize_weights(master_branch, auxiliary_branch) (label_net/helper.py:56-66); the backbone keeps ResNet's
        N(0, sqrt(2 / (k * k * out))) (backbone_resnet50.py:187-193)"""
        import math
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1.0)
                m.bias.data.zero_()
        for head in (self.master_branch, self.auxiliary_branch):
            for m in head.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight.data, nonlinearity='relu')
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.data.fill_(1.0)
                    m.bias.data.fill_(1e-4)

    def _logits(self, y, head_conv, size):
        z = run_conv(y, head_conv, None, None, out_f32=True)                      # [N, h, w, classes] f32
        return Fh.interpolate_bilinear(z, size).permute(0, 3, 1, 2)               # upstream's [N, classes, H, W]

    def forward(self, x):
        size = tuple(x.shape[2:])
        h = Fh.to_pm(x)
        stem = self.initial[0]
        h = run_conv(h, stem[0], stem[1], stem[2])
        h = run_conv(h, stem[3], stem[4], stem[5])
        h = run_conv(h, stem[6], self.initial[1], self.initial[2])
        h = Fh.max_pool2d(h, 3, 2, 1)
        f3 = self.layer3(self.layer2(self.layer1(h)))
        out = self._logits(self.master_branch[0](self.layer4(f3)), self.master_branch[1], size)
        if self.training and self.use_aux:
            a = self.auxiliary_branch
            y = Fh.dropout2d(run_conv(f3, a[0], a[1], a[2]), a[3].p, True)
            return [out, self._logits(y, a[4], size)]
        return out
