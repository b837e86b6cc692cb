"""CPU oracle (test infrastructure only) of the semantic-label net G_label: PSPNet over a dilated ResNet-50.

Plain torch.nn restatement, NCHW f32, state_dict keys identical to upstream's so that the same deterministic weights load
into upstream (fixture generation), this oracle and the HIP module:
  PSPNet / _PSPModule     lib/models/generator/label_net/labels_model.py:12-94
  ResNet / Bottleneck     lib/models/generator/label_net/backbone_resnet50.py:76-235 (resnet50(): deep stem, dilated,
                          multi-grid, :263-278)
Pinned by tests/golden/label_net_small.npz (generated from upstream's own class by tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):                                   # backbone_resnet50.py:76-123
    def __init__(self, cin, planes, stride, dilation, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride, bias=False), nn.BatchNorm2d(4 * planes))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


# (planes, stride of the first block, dilation of every block): _make_layer, backbone_resnet50.py:196-221, as resnet50() calls
# it -- layer3 'dilation=2' opens with a dilation-1 block, layer4 is the multi-grid (4, 8, 16)
STAGES = ((64, 1, (1, 1, 1)), (128, 2, (1, 1, 1, 1)), (256, 1, (1, 2, 2, 2, 2, 2)), (512, 1, (4, 8, 16)))


def make_stage(cin, planes, stride, dilations):
    blocks = []
    for i, d in enumerate(dilations):
        blocks.append(Bottleneck(cin if i == 0 else 4 * planes, planes, stride if i == 0 else 1, d,
                                 project=(i == 0 and (stride != 1 or cin != 4 * planes))))
    return nn.Sequential(*blocks)


class PyramidPooling(nn.Module):                               # _PSPModule, labels_model.py:12-40
    def __init__(self, cin, bins=(1, 2, 3, 6)):
        super().__init__()
        cout = cin // len(bins)
        self.stages = nn.ModuleList(nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Conv2d(cin, cout, 1, bias=False),
                                                  nn.BatchNorm2d(cout), nn.ReLU()) for b in bins)
        self.bottleneck = nn.Sequential(nn.Conv2d(cin + cout * len(bins), cout, 1), nn.BatchNorm2d(cout), nn.ReLU(),
                                        nn.Dropout2d(0.1))

    def forward(self, f):
        size = f.shape[2:]
        levels = [f] + [F.interpolate(st(f), size=size, mode='bilinear', align_corners=True) for st in self.stages]
        return self.bottleneck(torch.cat(levels, 1))


class PSPNet(nn.Module):                                       # labels_model.py:43-94
    def __init__(self, num_classes, use_aux=True):
        super().__init__()
        stem = nn.Sequential(nn.Conv2d(3, 64, 3, 2, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                             nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                             nn.Conv2d(64, 128, 3, 1, 1, bias=False))
        self.initial = nn.Sequential(stem, nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        cin = 128
        for i, (planes, stride, dil) in enumerate(STAGES, 1):
            setattr(self, f'layer{i}', make_stage(cin, planes, stride, dil))
            cin = 4 * planes
        self.use_aux = use_aux
        self.master_branch = nn.Sequential(PyramidPooling(cin), nn.Conv2d(cin // 4, num_classes, 1))
        self.auxiliary_branch = nn.Sequential(nn.Conv2d(cin // 2, cin // 4, 3, padding=1, bias=False), nn.BatchNorm2d(cin // 4),
                                              nn.ReLU(), nn.Dropout2d(0.1), nn.Conv2d(cin // 4, num_classes, 1))

    def forward(self, x):
        size = x.shape[2:]
        f3 = self.layer3(self.layer2(self.layer1(self.initial(x))))
        out = F.interpolate(self.master_branch(self.layer4(f3)), size=size, mode='bilinear', align_corners=True)
        if self.training and self.use_aux:
            aux = F.interpolate(self.auxiliary_branch(f3), size=size, mode='bilinear', align_corners=True)
            return [out, aux]
        return out
